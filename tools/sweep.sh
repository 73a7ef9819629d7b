#!/bin/bash
# parameter sweep of the wavefront trace kernel (GPU box)
run() { echo -n "$* : "; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run X=0
for t in 4 16 32 64; do run DDGI_WF_TAIL=$t; done
for f in 4 8 32 48; do run DDGI_WF_FETCH=$f; done
for p in 1024 1152 1280; do run DDGI_WF_POOL=$p; done
run DDGI_WF_TAIL=16 DDGI_WF_FETCH=8
run DDGI_WF_TAIL=32 DDGI_WF_FETCH=8
