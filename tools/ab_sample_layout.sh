#!/bin/bash
# A/B of the REF sampler's table layout (GPU box): every libddgi_probe*.so next to the package through tools/sample_bench.py (twice), then
# the kernels' times and FETCH_SIZE per library
D=$GRAFT_REPO_ROOT/dynamic-diffuse-global-illumination-minecraft_amd
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  echo "== $lib"; DDGI_LIB=$D/$lib python tools/sample_bench.py 2>/dev/null | grep "mode 0"
done; done
for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  tag=$(basename $lib .so); export DDGI_LIB=$D/$lib
  bash tools/pmc_sample.sh r05 $tag > /dev/null 2>&1
  echo "== $lib"; grep -E "k_probe_sample_ref|box_filter" gpurun_out/profiles_out/r05_${tag}_sample_kernels.txt | grep -E "avg_us|calls|FETCH|VALU | [0-9]+ +[0-9.]+ +[0-9.]+$"
done
