#!/bin/bash
# instruction-cache / fetch counters of the trace kernel for the bench workload (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DDGI_AQ_MARCH=${DDGI_AQ_MARCH:-5}   # every launch is the steady-state kernel (no wave-split measurement launches among them)
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march --no-extras"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o ic1 --output-format csv -- $BENCH > $OUT/ic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT -o ic2 --output-format csv -- $BENCH > $OUT/ic2.log 2>&1
python3 - <<PY
import csv, collections
for f in ["ic1","ic2"]:
    agg=collections.defaultdict(list)
    try:
        rows = list(csv.DictReader(open("$OUT/"+f+"_counter_collection.csv")))
    except Exception as e:
        print(f, "no data", e); continue
    for r in rows:
        if "trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(f,k,"%.4g"%(sum(v)/len(v)))
PY
