#!/bin/bash
# Round 6, second session — the last commit: smoke, the -m gpu suite as the driver runs it (-x), then the multi-process tests three more times, the default bench line
set -u
OUT=gpurun_out/r06_final3; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?: $(tail -1 $OUT/smoke.txt)" | tee -a $OUT/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 -p no:cacheprovider > $OUT/gputest.txt 2>&1
echo "gpu suite rc $? in $(( $(date +%s) - t0 )) s: $(grep -E "passed|failed" $OUT/gputest.txt | tail -1)" | tee -a $OUT/summary.txt
for k in 1 2 3; do
  t0=$(date +%s)
  timeout 900 python -m pytest tests/test_zz_gpu_exchange_p2p.py tests/test_zz_gpu_peer_loss.py tests/test_host_cpp.py -x -q -m gpu -p no:cacheprovider > $OUT/zz_$k.txt 2>&1
  echo "multi-process tests, round $k: rc $? in $(( $(date +%s) - t0 )) s: $(grep -E "passed|failed" $OUT/zz_$k.txt | tail -1)" | tee -a $OUT/summary.txt
done
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?: $(head -c 300 $OUT/bench_default.json)" | tee -a $OUT/summary.txt
