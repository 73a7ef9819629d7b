#!/bin/bash
# SQ PMC passes for the DDGI-mode blend kernel (GPU box); prints per-launch means
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_blend
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --mode ddgi --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU -d $OUT -o sq1 --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o mem --output-format csv -- $BENCH > $OUT/mem.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o memw --output-format csv -- $BENCH > $OUT/memw.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM -d $OUT -o sq2 --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES -d $OUT -o mfma --output-format csv -- $BENCH > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD -d $OUT -o lds --output-format csv -- $BENCH > $OUT/lds.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- $BENCH > $OUT/kt.log 2>&1
python3 - <<PY2
import csv, collections
agg = collections.OrderedDict()
try:
    for r in csv.DictReader(open("$OUT/kt_kernel_trace.csv")):
        if "blend" in r["Kernel_Name"]:
            agg.setdefault(r["Kernel_Name"].split("(")[0].replace("ddgi::", ""), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# kernel times (rocprofv3 --kernel-trace)")
    for k, v in agg.items(): print("%-28s calls %4d  avg %8.2f us  min %8.2f us" % (k[:28], len(v), sum(v) / len(v), min(v)))
except Exception as e:
    print("# kernel trace: no data", e)
PY2
python3 - <<PY
import csv, collections
print("# rocprofv3 --pmc passes on bench.py --mode ddgi --steps 3 --warmup 1: per-launch means of the DDGI blend kernels (C3: 16 384 probes x 256 rays)")
for f in ["sq1","sq2","mfma","lds","mem","memw"]:
    agg=collections.defaultdict(list)
    try:
        rows = list(csv.DictReader(open("$OUT/"+f+"_counter_collection.csv")))
    except Exception as e:
        print("#", f, "no data", e); continue
    for r in rows:
        if "blend" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("ddgi::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(agg.items()): print("%-28s %-24s %14.5g" % (k[:28], c, sum(v)/len(v)))
PY
