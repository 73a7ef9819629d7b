#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run X=0
run DDGI_AQ_POOL=1024
run DDGI_AQ_POOL=1280
run DDGI_AQ_POOL=1536
run DDGI_AQ_POOL=1792
run DDGI_AQ_POOL=2048
