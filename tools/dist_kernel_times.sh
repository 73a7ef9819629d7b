#!/bin/bash
# per-kernel times of bench.py's N > 1 code path run with one rank (GPU box): tools/dist_kernel_times.sh [bench args...]
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dist
DDGI_BENCH_FORCE_DIST=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_dist -o d --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2>&1
python $ROOT/tools/profile_summary.py /tmp/prof_dist/d "bench.py (DDGI_BENCH_FORCE_DIST=1) --steps 20 --warmup 5 $*" | grep -v "^#" | head -12
