#!/bin/bash
# issue counters of the trace kernel for one library build: tools/pmc_quick2.sh <tag> [lib suffix]   (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_q2_${1:-x}
mkdir -p $OUT
[ -n "$2" ] && export DDGI_LIB=$GRAFT_REPO_ROOT/dynamic-diffuse-global-illumination-minecraft_amd/libddgi_probe$2.so
cd /tmp && export TMPDIR=/tmp
export DDGI_AQ_MARCH=${DDGI_AQ_MARCH:-7} DDGI_FRAMES_IN_FLIGHT=1
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT -o ic2 --output-format csv -- $BENCH > $OUT/ic2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT -o ic1 --output-format csv -- $BENCH > $OUT/ic1.log 2>&1
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
    for k,v in sorted(agg.items()): print("$1", f,k,"%.4g"%(sum(v)/len(v)))
PY
