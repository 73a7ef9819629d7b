#!/bin/bash
# Round 6 (GPU box): (a) does hipIpcOpenMemHandle of a buffer of 2 GiB or more come back under the HIP / ROCr that PyTorch ships (bench.py's process) as it does under
# the system's (the tests' processes)?  (b) a larger ray pool (1472 slots, rings of 1728 entries: non-power-of-two modulo) against the default 1344 / 2048;
# (c) the blend kernels' counters.
set -u
OUT=gpurun_out/r06_probe2
mkdir -p $OUT gpurun_out/profiles_out
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
for mb in 1024 2047 2048 4096; do
  echo "== system ROCr, $mb MB, 2 processes" >> $OUT/ipc_2gib.txt
  ( cd tools/microbench && timeout 60 ./ipc_open_cost.bin 20 $mb 2 ) >> $OUT/ipc_2gib.txt 2>&1
  echo "== PyTorch's libamdhip64 / libhsa-runtime64 (LD_LIBRARY_PATH=$TL), $mb MB, 2 processes" >> $OUT/ipc_2gib.txt
  ( cd tools/microbench && LD_LIBRARY_PATH=$TL timeout 60 ./ipc_open_cost.bin 20 $mb 2 ) >> $OUT/ipc_2gib.txt 2>&1
  echo "   rc $?" >> $OUT/ipc_2gib.txt
done
( cd tools/microbench && LD_LIBRARY_PATH=$TL ldd ./ipc_open_cost.bin | grep -i "hip\|hsa" ) >> $OUT/ipc_2gib.txt 2>&1
cat $OUT/ipc_2gib.txt | tee -a $OUT/summary.txt
for rep in 1 2; do for lib in libddgi_probe.so libddgi_probe_pool1472.so; do
  for cfg in "--workload c3" "--workload c3 --mode ddgi"; do
    echo -n "$lib $cfg: " >> $OUT/pool_ab.txt
    DDGI_LIB=$D/$lib python bench.py $cfg --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.4f ms/step  kernel %.4f  mw %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('tuning',{}).get('march_waves')))" >> $OUT/pool_ab.txt
  done
done; done
DDGI_LIB=$D/libddgi_probe_pool1472.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -x -p no:cacheprovider > $OUT/pool1472_parity.txt 2>&1
echo "parity on the 1472-slot pool: rc $?: $(tail -1 $OUT/pool1472_parity.txt)" >> $OUT/pool_ab.txt
cat $OUT/pool_ab.txt | tee -a $OUT/summary.txt
bash tools/pmc_blend.sh > $OUT/pmc_blend.txt 2>&1
cat $OUT/pmc_blend.txt >> $OUT/summary.txt
