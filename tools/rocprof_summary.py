#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (gpurun_out/prof/*_results.db, from
`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`) into the plain-text per-kernel summary
committed under profiles/.  Usage: tools/rocprof_summary.py results.db > profiles/rNN_name.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}")
    print("# times in microseconds")
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>14s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>7s}")
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        print(f"{name[:60]:60s} {n:6d} {s / 1e3:14.3f} {a / 1e3:12.3f} {mn / 1e3:12.3f} {mx / 1e3:12.3f} {100.0 * s / tot:7.3f}")
    print()
    print("# launch geometry / resources per kernel")
    for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels group by name"):
        print("%s: grid=%d wg=%d lds=%dB arch_vgpr=%d accum_vgpr=%d sgpr=%d scratch=%dB" % r)


if __name__ == "__main__":
    main(sys.argv[1])
