#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run X=0
run DDGI_TRACE_KERNEL=async
run DDGI_TRACE_KERNEL=async DDGI_AQ_MARCH=8
run DDGI_TRACE_KERNEL=async DDGI_WF_FETCH=4
run DDGI_TRACE_KERNEL=async DDGI_WF_FETCH=24
