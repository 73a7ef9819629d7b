#!/bin/bash
# parameter sweep of the wavefront trace kernel (GPU box)
run() { echo -n "$* : "; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run X=0
for t in 8 12 24 32; do run DDGI_WF_TAIL=$t; done
for f in 8 24 32; do run DDGI_WF_FETCH=$f; done
run DDGI_WF_TAIL=24 DDGI_WF_FETCH=24
