#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run DDGI_AQ_MARCH=5
run DDGI_AQ_MARCH=5 DDGI_AQ_POOL=1024
run DDGI_AQ_MARCH=5 DDGI_AQ_POOL=1280
run DDGI_AQ_MARCH=5 DDGI_AQ_POOL=2048
run DDGI_AQ_MARCH=5 DDGI_WF_FETCH=4
run DDGI_AQ_MARCH=5 DDGI_WF_FETCH=16
run DDGI_AQ_MARCH=5 DDGI_WF_FETCH=32
run DDGI_AQ_MARCH=5
