#!/bin/bash
# A/B of run-time tuning on one box:  tools/env_ab.sh "VAR=a" "VAR=b" ...   (each argument: environment assignments for one variant)
run() { echo -n "$1 : "; env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d.get('fast_march',{}); print('%.3f ms  (march waves %s)   fast %.3f ms' % (d['roofline']['kernel_ms'], d['tuning']['march_waves'], f.get('kernel_ms', float('nan'))))"; }
for i in 1 2; do for v in "$@"; do run "$v"; done; done
