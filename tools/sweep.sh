#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 8 --no-cpu-baseline 2>&1 | grep -E "^\{|ddgi\]" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('%.3f ms' % d['roofline']['kernel_ms'])
    else: print(l.strip(), end=' ')"; }
run DDGI_VERBOSE=1
run DDGI_AQ_MARCH=5
run DDGI_AQ_MARCH=6
run DDGI_VERBOSE=1
run DDGI_AQ_MARCH=5
