#!/bin/bash
# parameter sweep of the wavefront trace kernel (GPU box)
run() { echo -n "$* : "; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run X=0
run DDGI_WF_THREADS=512
run DDGI_WF_THREADS=512 DDGI_WF_TAIL=8
run DDGI_WF_THREADS=512 DDGI_WF_TAIL=32
run DDGI_WF_THREADS=512 DDGI_WF_FETCH=8
