#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run X=0
run DDGI_WF_TAIL=2
run DDGI_WF_FETCH=4
run DDGI_WF_FETCH=16
run DDGI_WF_FETCH=24
run DDGI_WF_FETCH=32
