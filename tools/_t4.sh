cd $GRAFT_REPO_ROOT
AB_ARGS="--no-fast-march" bash tools/ab.sh withopt 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r3g_tests.log 2>&1; grep -E 'passed|failed|error' gpurun_out/r3g_tests.log | tail -3
