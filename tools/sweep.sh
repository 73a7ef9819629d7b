#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
for i in 1 2; do
run DDGI_LUT_OFF=0
run DDGI_LUT_OFF=1
run DDGI_LUT_OFF=2
run DDGI_LUT_OFF=3
done
