#!/bin/bash
# kernel times + SQ counters of the cage-sample kernels on tools/sample_bench.py (GPU box) -> gpurun_out/profiles_out/<round>_<tag>_sample_kernels.txt
R=${1:-r03}; T=${2:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sample_$T; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/profiles_out
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/sample_bench.py"
rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -d $OUT -o sq1 --output-format csv -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $OUT -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o mem --output-format csv -- $CMD > $OUT/mem.log 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/profiles_out/${R}_${T}_sample_kernels.txt
import csv, collections
print("# tools/sample_bench.py: 1 440 000 scattered shading points over the C3 grid, REF then DDGI mode; rocprofv3 --kernel-trace and --pmc passes")
rows = list(csv.DictReader(open("$OUT/kt_kernel_trace.csv")))
agg = collections.OrderedDict()
for r in rows:
    if "sample" in r["Kernel_Name"]:
        agg.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-48s %6s %10s %10s" % ("kernel", "calls", "avg_us", "min_us"))
for k, v in agg.items(): print("%-48s %6d %10.2f %10.2f" % (k[:48], len(v), sum(v) / len(v), min(v)))
for f in ("sq1", "sq2", "mem"):
    a = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open("$OUT/" + f + "_counter_collection.csv")):
            if "sample" in r["Kernel_Name"]: a[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    except Exception as e:
        print("#", f, "no data:", e); continue
    for (k, c), v in sorted(a.items()): print("%-40s %-26s %14.5g" % (k[:40], c, sum(v) / len(v)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/profiles_out/${R}_${T}_sample_kernels.txt
