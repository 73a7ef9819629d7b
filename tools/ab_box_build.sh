#!/bin/bash
# the REF sampler's table build (k_sample_box_filter) per library on C3 and C4 (GPU box): kernel times under rocprofv3 + the bench's first batch after an update
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  echo "== $lib"
  DDGI_LIB=$D/$lib bash tools/sample_kernel_times.sh bb_$lib 2>/dev/null | grep -E "box_filter|sample_ref"
  DDGI_LIB=$D/$lib python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sample']; print('c4: first batch after an update %.3f ms, steady %.3f ms' % (s['first_batch_after_update_ms'], s['ms']))"
done
