#!/bin/bash
# LDS bank conflicts of every kernel of a command (GPU box): conflict cycles / LDS-active cycles, with the kernels' times
#   tools/pmc_lds_conflicts.sh <tag> <command...>
T=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $OUT -o c --output-format csv -- "$@" > $OUT/c.log 2>&1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/c_counter_collection.csv")):
    agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-60s %8s %12s %12s %8s %12s" % ("kernel", "launches", "conflict", "lds_active", "ratio", "wait_lds/wave_cycles"))
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    a = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print("%-60s %8d %12.4g %12.4g %8.2f %12.3f" % (k, len(c["SQ_WAVE_CYCLES"]), m.get("SQ_LDS_BANK_CONFLICT", 0), a, m.get("SQ_LDS_BANK_CONFLICT", 0) / a if a else 0, m.get("SQ_WAIT_INST_LDS", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1))))
PY
