cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r3e/gputest.log
timeout 400 python bench.py --workload c5 --mode ddgi --steps 12 > gpurun_out/r3e/c5_sdyn.json 2> gpurun_out/r3e/c5_err.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench_err.log
tail -3 gpurun_out/r3e/gputest.log; cat gpurun_out/r3e/c5_sdyn.json | cut -c1-300
