#!/bin/bash
# Round 6 (GPU box): event rings per (bucket, update parity) of k_probe_trace_aq against the default build.
#   libddgi_probe_staging.so  -DDDGI_AQ_STAGING=1: 128-entry staging rings in front of the buckets' rings (semaphore per ring)
#   libddgi_probe_parity3.so  -DDDGI_AQ_STAGING=3 -DDDGI_AQ_POOL_CT=1216: the event rings themselves per (bucket, parity), pool-deep, 128 slots fewer
# bit-exactness first, then one rank's DDGI slab and the whole grid, then REF (what the smaller pool costs where nothing mixes).
set -u
OUT=gpurun_out/ab_staging
mkdir -p $OUT
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
LIBS=$(cd $D; ls libddgi_probe*.so | grep -v prof)
for lib in $LIBS; do
  [ $lib = libddgi_probe.so ] && continue
  DDGI_LIB=$D/$lib timeout 900 python -m pytest tests/test_gpu_ddgi_frames_in_flight.py tests/test_gpu_ddgi_mode.py tests/test_gpu_frames_in_flight.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > $OUT/parity_$lib.txt 2>&1
  echo "parity on $lib: rc $?: $(grep -E 'passed|failed' $OUT/parity_$lib.txt | tail -1)" | tee -a $OUT/summary.txt
done
for rep in 1 2; do for lib in $LIBS; do
  echo "== $lib (rep $rep)" >> $OUT/summary.txt
  DDGI_LIB=$D/$lib FIF_MODE=ddgi FIF_WORLDS=1,4,8 FIF_FIFS=1,8 timeout 300 python tools/fif_timing.py 2>/dev/null | grep world | cut -c1-110 >> $OUT/summary.txt
  DDGI_LIB=$D/$lib FIF_WORLDS=1,8 FIF_FIFS=8 timeout 300 python tools/fif_timing.py 2>/dev/null | grep world | cut -c1-110 >> $OUT/summary.txt
done; done
cat $OUT/summary.txt
