#!/bin/bash
# C5 / S-Dyn per library (GPU box): ms per frame and the event-timed blend; twice
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for rep in 1 2; do for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  echo -n "$lib: "; DDGI_LIB=$D/$lib timeout 400 python bench.py --workload c5 --mode ddgi --steps 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per frame, blend %.4f ms' % (d['ms_per_step'], d['blend']['kernel_ms']))"
done; done
