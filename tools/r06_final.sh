#!/bin/bash
# Round 6, final checks on the GPU box: smoke, the whole -m gpu suite as the driver runs it, the big-ring probe, one more bench line on the round's last build
set -u
OUT=gpurun_out/r06_final
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?: $(tail -1 $OUT/smoke.txt)" | tee -a $OUT/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 -p no:cacheprovider > $OUT/gputest.txt 2>&1
echo "gpu suite rc $? in $(( $(date +%s) - t0 )) s: $(grep -E "passed|failed" $OUT/gputest.txt | tail -1)" | tee -a $OUT/summary.txt
( cd tools/microbench && timeout 400 ./ipc_big_ring_probe.bin 2048 0 1 2 4 8 16 32 63 ) > $OUT/ipc_big_ring_probe.txt 2>&1
cat $OUT/ipc_big_ring_probe.txt | tee -a $OUT/summary.txt
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?: $(head -c 300 $OUT/bench_default.json)" | tee -a $OUT/summary.txt
