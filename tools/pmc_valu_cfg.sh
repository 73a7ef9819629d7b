#!/bin/bash
# VALU / SALU wave-instructions, VALU-active cycles and wait cycles per launch of the trace kernel for a library and a pinned wave split (GPU box):
#   tools/pmc_valu_cfg.sh <library suffix, "" for the release build> <march waves>
D=$GRAFT_REPO_ROOT/dynamic-diffuse-global-illumination-minecraft_amd
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_cfg
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
DDGI_LIB=$D/libddgi_probe$1.so DDGI_AQ_MARCH=$2 DDGI_FRAMES_IN_FLIGHT=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT -o v --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march --no-extras > $OUT/v.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/v_counter_collection.csv")):
    if "trace_aq" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("lib '$1' march waves $2: " + " ".join("%s=%.4g" % (k, sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
