#!/bin/bash
# profiling ablations of the trace kernel on the GPU box (results are NOT exact when DDGI_ABLATE != 0)
for cfg in "0 16" "0 1" "0 32" "0 48" "1 16" "2 16" "3 16"; do
  set -- $cfg
  echo -n "ablate=$1 wait=$2 : "
  DDGI_ABLATE=$1 DDGI_WAIT_THRESHOLD=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms')"
done
