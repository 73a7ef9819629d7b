#!/usr/bin/env python3
"""Turns the counter passes of tools/pmc_icache.sh (gpurun_out/pmc_icache_<tag>/) into the `*_issue.txt` summary that
bench.py reads for `roofline.issue`:  tools/pmc_issue.py <tag> <out.txt>"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, out):
    d = os.path.join(ROOT, "gpurun_out", "pmc_icache_" + tag)
    c = {}
    for f in ("ic1", "ic2"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(d, f + "_counter_collection.csv"))):
            if "trace" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        c.update({k: sum(v) / len(v) for k, v in agg.items()})
    lines = [f"# rocprofv3 --pmc passes (tools/pmc_icache.sh {tag}), bench workload c3_cave_32x16x32_probes_x256_rays_ref",
             "# per-launch means over the k_probe_trace_aq dispatches of `bench.py --steps 3 --warmup 1 --no-fast-march` with the march/event",
             "# wave split pinned (DDGI_AQ_MARCH=%s)%s" % (os.environ.get("DDGI_AQ_MARCH", "5"), "; DDGI_FAST_MARCH=1: the tolerance-mode kernel" if os.environ.get("DDGI_FAST_MARCH") == "1" else "")]
    for k in sorted(c):
        lines.append(f"{k:28s} {c[k]:.6g}")
    busy = c["SQ_ACTIVE_INST_VALU"] / (c["SQ_WAVE_CYCLES"] / 4.0)
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    lines.append(f"# instruction cache: {c['SQC_ICACHE_MISSES']:.0f} misses in {c['SQC_ICACHE_REQ']:.3g} requests")
    lines.append(f"# VALU busy = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 4 waves per SIMD) = {busy:.3f}")
    lines.append(f"# mean active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {lanes:.3f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:])
