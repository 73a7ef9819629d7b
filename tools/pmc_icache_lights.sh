#!/bin/bash
# instruction-cache counters of the one-light and of the four-light trace kernel (C3-sized grid, DDGI mode; tools/lights_timing.py --one)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache_lights
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DDGI_AQ_MARCH=${DDGI_AQ_MARCH:-7}
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o ic1 --output-format csv -- python $GRAFT_REPO_ROOT/tools/lights_timing.py --one > $OUT/ic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT -o ic2 --output-format csv -- python $GRAFT_REPO_ROOT/tools/lights_timing.py --one > $OUT/ic2.log 2>&1
python3 - <<PY
import csv, collections
for f in ["ic1","ic2"]:
    agg=collections.defaultdict(list)
    try:
        rows = list(csv.DictReader(open("$OUT/"+f+"_counter_collection.csv")))
    except Exception as e:
        print(f, "no data", e); continue
    for r in rows:
        if "trace_aq" in r["Kernel_Name"]:
            agg[("4 lights" if "CfgMulti" in r["Kernel_Name"] else "1 light ", r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(f,k[0],k[1],"%.4g"%(sum(v)/len(v)), "(%d launches)"%len(v))
PY
