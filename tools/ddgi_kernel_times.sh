#!/bin/bash
# per-kernel times of the DDGI-mode bench (GPU box): tools/ddgi_kernel_times.sh [extra env assignments...]
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dk
env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_dk -o ddgi --output-format csv -- python $ROOT/bench.py --mode ddgi --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/profile_summary.py /tmp/prof_dk/ddgi "bench.py --mode ddgi --steps 20 --warmup 5 $*" | grep -v "^#" | head -12
