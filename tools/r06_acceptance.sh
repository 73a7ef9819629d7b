#!/bin/bash
# Round 6 acceptance run (GPU box): the whole -m gpu suite in its new order with durations, the C5 DDGI bring-up at 4 ranks on one GPU (ranks map their
# peers in turns), the record pointer's laundering A/B in DDGI mode, the samplers' HBM counters.
set -u
OUT=gpurun_out/r06_accept
mkdir -p $OUT gpurun_out/profiles_out
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x --durations=45 -p no:cacheprovider > $OUT/gputest.txt 2>&1
echo "gpu suite rc $? in $(( $(date +%s) - t0 )) s: $(tail -1 $OUT/gputest.txt)" | tee -a $OUT/summary.txt
DDGI_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 \
    --workload c5 --mode ddgi --exchange p2p --p2p-timeout 60 --steps 4 --warmup 2 > $OUT/c5_ddgi_4ranks_one_gpu.json 2> $OUT/c5_ddgi_4ranks.err
echo "c5 ddgi 4 ranks rc $?: $(head -c 1500 $OUT/c5_ddgi_4ranks_one_gpu.json)" | tee -a $OUT/summary.txt
tail -3 $OUT/c5_ddgi_4ranks.err | cut -c1-400 >> $OUT/summary.txt
for rep in 1 2; do for lib in libddgi_probe.so libddgi_probe_nolaunder.so; do
  for cfg in "--workload c3 --mode ddgi" "--workload c3"; do
    echo -n "$lib $cfg: " >> $OUT/launder_ab.txt
    DDGI_LIB=$D/$lib python bench.py $cfg --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.4f ms/step  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" >> $OUT/launder_ab.txt
  done
done; done
cat $OUT/launder_ab.txt >> $OUT/summary.txt
python tools/pmc_sample_traffic.py r06 a > $OUT/sample_traffic.txt 2>&1
cat $OUT/sample_traffic.txt >> $OUT/summary.txt
cat $OUT/summary.txt
