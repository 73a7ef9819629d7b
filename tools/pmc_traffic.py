#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes collected by tools/pmc_run.sh (gpurun_out/pmc_<tag>/) into
  profiles/<round>_pmc_<tag>.txt           per-launch counter means of the trace kernel
  profiles/<round>_traffic_<tag>.json      HBM bytes per launch (read by bench.py -> roofline.traffic)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  MI355X_MICROARCH.md (HBM section): on
gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced stream at 64 B, i.e. it reads HALF
the bytes of such a stream -> the read side is doubled ("corrected"); WRITE_SIZE is taken as is."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def means(path, kernel_substr="trace"):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main(tag, rnd="r01", workload="c3_cave_32x16x32_probes_x256_rays_ref"):
    d = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
    allc = {}
    for f in ("sq1", "sq2", "fetch", "write", "grbm"):
        allc.update(means(os.path.join(d, f + "_counter_collection.csv")))
    lines = [f"# rocprofv3 --pmc passes (tools/pmc_run.sh {tag}), bench workload {workload}",
             "# per-launch means over the k_probe_trace_* dispatches of `bench.py --steps 3 --warmup 1`"]
    for k in sorted(allc):
        lines.append(f"{k:28s} {allc[k]:.6g}")
    fetch_b = allc.get("FETCH_SIZE", 0.0) * 1024.0
    write_b = allc.get("WRITE_SIZE", 0.0) * 1024.0
    if "GRBM_GUI_ACTIVE" in allc:
        cyc = allc["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        lines.append(f"# kernel duration ~ GRBM_GUI_ACTIVE/8 = {cyc:.4g} cycles")
        if "SQ_ACTIVE_INST_VALU" in allc:
            lines.append(f"# VALU busy ~ SQ_ACTIVE_INST_VALU*4/(1024 SIMDs)/cycles = {allc['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:.3f}")
    lines.append(f"# HBM read  : FETCH_SIZE {fetch_b / 1e6:.1f} MB raw, {2 * fetch_b / 1e6:.1f} MB with the gfx950 x2 correction")
    lines.append(f"# HBM write : WRITE_SIZE {write_b / 1e6:.1f} MB")
    outdir = os.path.join(ROOT, "gpurun_out", "profiles_out")  # merged back by gpurun; copy what is to be kept into profiles/
    os.makedirs(outdir, exist_ok=True)
    txt = os.path.join(outdir, f"{rnd}_pmc_{tag}.txt")
    open(txt, "w").write("\n".join(lines) + "\n")
    js = os.path.join(outdir, f"{rnd}_traffic_{tag}.json")
    json.dump({"workload": workload, "kernel": "k_probe_trace_aq", "fetch_size_bytes_raw": fetch_b,
               "fetch_size_bytes_corrected": 2 * fetch_b, "write_size_bytes": write_b,
               "hbm_bytes_per_launch": 2 * fetch_b + write_b,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read side doubled per MI355X_MICROARCH.md"},
              open(js, "w"), indent=1)
    print(open(txt).read())


if __name__ == "__main__":
    main(*sys.argv[1:])
