#!/usr/bin/env python3
"""Where k_probe_trace_aq's VALU wave-instructions go (C3, REF, the headline instantiation), from three sources:

  static   the kernel's assembly, compiled with -DDDGI_MARKS=1 (comment lines delimit the march waves' trip: fetch | burst | finish);
           instructions are counted per region by kind
  dynamic  the counters build (tools/aq_stats.py): march bursts and event groups per update
  measured rocprofv3 --pmc SQ_INSTS_VALU per launch (tools/pmc_icache.sh -> profiles/*_issue.txt)

    tools/valu_attribution.py --bursts 965000 --groups 670000 --valu 1.11068e9 [--asm marks.s]

A march trip executes its fetch region at most once, one of the burst's two variants (with / without the iteration-limit test)
and the finish region once: bursts x static count is exact for the burst's steps and an upper bound for fetch and finish (their
conditional parts).  What is left of the measured total is the event waves' (polling, claims, the events themselves)."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN4ddgi16k_probe_trace_aqILb0ELi1344ENS_8CfgPlainILi0ELb0EEEEEvNS_9TraceArgsEiiNS_7AqChainEPj"


def compile_marked(out):
    src = os.path.join(ROOT, "dynamic-diffuse-global-illumination-minecraft_amd", "csrc", "ddgi_trace_wf.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mllvm",
           "-amdgpu-atomic-optimizer-strategy=None", "-DDDGI_MARKS=1", "-x", "hip", "-c", src, "-S", "--cuda-device-only", "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def kinds(lines):
    c = collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith((".", ";", "_")) or l.endswith(":"):
            continue
        op = l.split()[0]
        c["all"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
            if op.startswith(("v_readlane", "v_writelane")):
                c["spill_lane_moves"] += 1
            if op.startswith("v_lshl_add_u64") or op.startswith(("v_add_co", "v_addc_co", "v_mad_u64")):
                c["address_arithmetic"] += 1
            if op.startswith("v_pk_"):
                c["packed"] += 1
            if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_mul_lo", "v_mul_hi", "v_sin", "v_cos", "v_exp", "v_log")):
                c["quarter_rate"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "scratch_", "flat_", "buffer_")):
            c["vmem"] += 1
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None)
    ap.add_argument("--bursts", type=float, required=True, help="march bursts (wave trips) per update, counters build")
    ap.add_argument("--groups", type=float, required=True, help="event groups per update, counters build")
    ap.add_argument("--valu", type=float, required=True, help="SQ_INSTS_VALU per launch")
    ap.add_argument("--steps", type=int, default=24)
    a = ap.parse_args()
    path = a.asm or "/tmp/ddgi_marks.s"
    if not a.asm:
        compile_marked(path)
    s = open(path).read()
    i = s.index(KERNEL + ":")
    body = s[i:s.index(".Lfunc_end", i)].split("\n")
    mark = {}
    for n, l in enumerate(body):
        m = re.search(r"DDGI_MARK (\w+)", l)
        if m:
            mark[m.group(1)] = n
    fetch = kinds(body[mark["march_trip_begin"]:mark["march_burst_begin"]])
    burst = kinds(body[mark["march_burst_begin"]:mark["march_burst_end"]])
    finish = kinds(body[mark["march_burst_end"]:mark["march_trip_end"]])
    whole = kinds(body)
    rest = collections.Counter(whole)
    for c in (fetch, burst, finish):
        rest.subtract(c)
    per_variant = burst["valu"] / 2.0   # two unrolled variants of the burst, one runs
    rows = [
        ("march: burst steps (%d steps x %.1f VALU, one of two unrolled variants)" % (a.steps, per_variant / a.steps), a.bursts * per_variant, "exact"),
        ("march: fetch (claim, ring entry, 9 LDS words, 3 reciprocals, position)", a.bursts * fetch["valu"], "upper bound: once per trip"),
        ("march: finish (occupancy, escape test, hit class, state out, event-queue push)", a.bursts * finish["valu"], "upper bound: once per trip"),
    ]
    march = sum(r[1] for r in rows)
    rows.append(("event waves: polls, claims, events, pushes (the measured total minus the above)", a.valu - march, "per event group: %.0f" % ((a.valu - march) / a.groups)))
    print("# k_probe_trace_aq<false, 1344, CfgPlain<0>>, C3 REF: VALU wave-instructions per update by section")
    print("# measured SQ_INSTS_VALU %.4g; march bursts %.4g, event groups %.4g per update (counters build)" % (a.valu, a.bursts, a.groups))
    print("%-92s %12s %7s  %s" % ("section", "VALU", "share", "how"))
    for name, v, how in rows:
        print("%-92s %12.4g %6.1f%%  %s" % (name, v, 100.0 * v / a.valu, how))
    print("%-92s %12.4g %6.1f%%" % ("sum", sum(r[1] for r in rows), 100.0 * sum(r[1] for r in rows) / a.valu))
    print("#\n# static instruction mix of the kernel by region (the whole kernel: %d instructions, %d VALU)" % (whole["all"], whole["valu"]))
    print("%-34s %6s %6s %6s %6s %6s | %s" % ("region", "VALU", "SALU", "LDS", "VMEM", "all", "of the VALU: v_readlane/v_writelane (SGPR spills), 64-bit address arithmetic, packed, quarter-rate"))
    for name, c in (("march fetch", fetch), ("march burst (both variants)", burst), ("march finish", finish), ("everything else (events, set-up)", rest)):
        print("%-34s %6d %6d %6d %6d %6d | %5d %5d %5d %5d" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], c["all"], c["spill_lane_moves"], c["address_arithmetic"], c["packed"], c["quarter_rate"]))


if __name__ == "__main__":
    main()
