#!/bin/bash
# bench every libddgi_probe*.so build found next to the package (A/B of compile-time variants), twice
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for rep in 1 2; do
for lib in $(cd $D; ls libddgi_probe*.so); do
  echo -n "$lib $* : "; env DDGI_LIB=$D/$lib "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"
done; done
