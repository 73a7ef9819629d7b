#!/bin/bash
# Round 6 (GPU box): the staging rings per (bucket, update parity) of k_probe_trace_aq's event queues (-DDDGI_AQ_STAGING=1: libddgi_probe_staging.so) against the default:
# bit-exactness first, then one rank's DDGI slab and the whole grid at frames in flight 8, then the bench lines.
set -u
OUT=gpurun_out/ab_staging
mkdir -p $OUT
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
DDGI_LIB=$D/libddgi_probe_staging.so timeout 900 python -m pytest tests/test_gpu_ddgi_frames_in_flight.py tests/test_gpu_ddgi_mode.py tests/test_gpu_frames_in_flight.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > $OUT/parity_staging.txt 2>&1
echo "parity on the staging build: rc $?: $(grep -E 'passed|failed' $OUT/parity_staging.txt | tail -1)" | tee -a $OUT/summary.txt
for rep in 1 2; do for lib in libddgi_probe.so libddgi_probe_staging.so; do
  echo "== $lib (rep $rep)" >> $OUT/summary.txt
  DDGI_LIB=$D/$lib FIF_MODE=ddgi FIF_WORLDS=1,4,8 FIF_FIFS=1,8 timeout 300 python tools/fif_timing.py 2>/dev/null | grep world | cut -c1-110 >> $OUT/summary.txt
  DDGI_LIB=$D/$lib FIF_WORLDS=1,8 FIF_FIFS=8 timeout 300 python tools/fif_timing.py 2>/dev/null | grep world | cut -c1-110 >> $OUT/summary.txt
done; done
cat $OUT/summary.txt
