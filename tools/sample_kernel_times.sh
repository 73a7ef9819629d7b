#!/bin/bash
# kernel times of tools/sample_bench.py (REF then DDGI mode): tools/sample_kernel_times.sh <tag>   (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/sample_kt_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/sample_bench.py > $OUT/log.txt 2>&1
grep mode $OUT/log.txt
python $GRAFT_REPO_ROOT/tools/profile_summary.py $OUT/s "tools/sample_bench.py" | grep -E "sample|box|calls"
