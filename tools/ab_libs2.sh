#!/bin/bash
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for lib in $(cd $D; ls libddgi_probe*.so); do
for m in 2 3 4 5; do
  echo -n "$lib march=$m : "; env DDGI_LIB=$D/$lib DDGI_AQ_MARCH=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"
done; done
