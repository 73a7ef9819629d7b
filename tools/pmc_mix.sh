#!/bin/bash
# VALU instruction mix of the trace kernel for the bench workload (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march"
export DDGI_AQ_MARCH=5
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 -d $OUT -o m1 --output-format csv -- $BENCH > $OUT/m1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT -o m2 --output-format csv -- $BENCH > $OUT/m2.log 2>&1
python3 - <<PY
import csv, collections
for f in ["m1","m2"]:
    agg=collections.defaultdict(list)
    try:
        rows=list(csv.DictReader(open("$OUT/"+f+"_counter_collection.csv")))
    except Exception as e:
        print(f,"no data",e); continue
    for r in rows:
        if "trace" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(f,k,"%.4g"%(sum(v)/len(v)))
PY
