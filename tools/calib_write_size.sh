#!/bin/bash
# WRITE_SIZE / FETCH_SIZE against a known store pattern (tools/microbench/write_size_calib.hip) -> gpurun_out/profiles_out/<round>_write_size_calibration.txt
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/profiles_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o w --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/write_size_calib.bin > $OUT/w.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT -o wr --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/write_size_calib.bin > $OUT/wr.log 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/profiles_out/${R}_write_size_calibration.txt
import csv, collections
print("# rocprofv3 --pmc WRITE_SIZE (KiB) on tools/microbench/write_size_calib.bin: each kernel stores 4 194 304 dwords = 16.777 MB of payload")
for f in ("w", "wr"):
    agg = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open("$OUT/" + f + "_counter_collection.csv")):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    except Exception as e:
        print("#", f, "no data:", e); continue
    for (k, c), v in sorted(agg.items()):
        m = sum(v) / len(v)
        extra = "  = %.1f MB = %.2f x the payload" % (m * 1024 / 1e6, m * 1024 / 16777216.0) if c == "WRITE_SIZE" else ""
        print("%-16s %-24s %14.1f%s" % (k, c, m, extra))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/profiles_out/${R}_write_size_calibration.txt
