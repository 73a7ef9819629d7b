#!/bin/bash
# SQ_INSTS_VALU / SALU per launch of the trace kernel under a given env (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT -o v --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/v.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/v_counter_collection.csv")):
    if "trace" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(" ".join("%s=%.4g" % (k, sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
