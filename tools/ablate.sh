#!/bin/bash
# profiling ablations of the trace kernel on the GPU box (results are NOT exact when DDGI_ABLATE != 0)
run() { echo -n "$* : "; env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
run DDGI_ABLATE=0
run DDGI_ABLATE=1
run DDGI_ABLATE=2
run DDGI_ABLATE=3
run DDGI_ABLATE=4
run DDGI_NO_NOISE_LUT=1
