#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the trace kernel for a given env (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fw
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march"
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d $OUT -o $c --output-format csv -- $BENCH > $OUT/$c.log 2>&1
python3 - <<PY
import csv
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$OUT/${c}_counter_collection.csv")) if "trace" in r["Kernel_Name"]]
print("$c", "%.1f MB" % (sum(v)/len(v)*1024/1e6))
PY
done
