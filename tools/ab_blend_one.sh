#!/bin/bash
# the blend as one launch against two (GPU box): event-timed blend of the DDGI bench line + per-kernel times, per library, twice
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for rep in 1 2; do for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  echo "== $lib"
  DDGI_LIB=$D/$lib python bench.py --mode ddgi --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.4f' % d['ms_per_step'], json.dumps(d.get('blend')))"
  bash tools/ddgi_kernel_times.sh DDGI_LIB=$D/$lib | grep -i "blend"
done; done
