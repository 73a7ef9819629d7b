#!/bin/bash
# Round 6 (GPU box): staging variant 2 (a bucket's own ring before its staging rings) against the default and variant 1
set -u
OUT=gpurun_out/ab_staging2
mkdir -p $OUT
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
DDGI_LIB=$D/libddgi_probe_staging2.so timeout 600 python -m pytest tests/test_gpu_ddgi_frames_in_flight.py -q -m gpu -x -p no:cacheprovider > $OUT/parity_staging2.txt 2>&1
echo "parity on staging variant 2: rc $?: $(grep -E 'passed|failed' $OUT/parity_staging2.txt | tail -1)" | tee -a $OUT/summary.txt
for rep in 1 2; do for lib in libddgi_probe.so libddgi_probe_staging.so libddgi_probe_staging2.so; do
  echo "== $lib (rep $rep)" >> $OUT/summary.txt
  DDGI_LIB=$D/$lib FIF_MODE=ddgi FIF_WORLDS=1,8 FIF_FIFS=8 timeout 300 python tools/fif_timing.py 2>/dev/null | grep world | cut -c1-110 >> $OUT/summary.txt
done; done
cat $OUT/summary.txt
