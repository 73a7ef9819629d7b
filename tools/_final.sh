cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E 'RCCL|HIP version|ROCm version|Hostname|Librccl' | tail -8 > gpurun_out/r3h/gputest.log
tail -3 gpurun_out/r3h/gputest.log
bash tools/collect_profiles.sh r03 zd > gpurun_out/collect_zd.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/profiles_out/r03_zd_bench.json')); print('bench', d['ms_per_step'], d['roofline']['kernel_ms'], d['cpu_baseline']['parity_checked'])"
