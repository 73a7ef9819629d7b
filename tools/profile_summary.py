#!/usr/bin/env python3
"""Plain-text per-kernel summary of a `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o NAME` run:
  tools/profile_summary.py DIR/NAME [note ...] > profiles/rNN_name_kernel_stats.txt
(reads NAME_kernel_trace.csv: per-dispatch start/end timestamps and launch resources)."""
import collections
import csv
import sys


def main(prefix, note=""):
    rows = list(csv.DictReader(open(prefix + "_kernel_trace.csv")))
    agg = collections.OrderedDict()
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(r["Kernel_Name"], {"t": [], "r": r})
        a["t"].append(d)
    tot = sum(sum(a["t"]) for a in agg.values()) or 1.0
    if note:
        print("# " + note)
    print("# rocprofv3 --kernel-trace --stats summary of %s_kernel_trace.csv; times in microseconds" % prefix.split("/")[-1])
    print(f"{'kernel':64s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
    for name, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["t"])):
        t = a["t"]
        print(f"{name[:64]:64s} {len(t):6d} {sum(t):12.3f} {sum(t) / len(t):10.3f} {min(t):10.3f} {max(t):10.3f} {100.0 * sum(t) / tot:7.3f}")
    print("\n# launch geometry / resources per kernel")
    for name, a in agg.items():
        r = a["r"]
        print("%s: grid=%s wg=%s lds=%sB arch_vgpr=%s accum_vgpr=%s sgpr=%s scratch=%sB" % (
            name[:90], r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"]))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
