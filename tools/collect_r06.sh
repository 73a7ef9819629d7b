#!/bin/bash
# Round 6's evidence, collected on the GPU box into gpurun_out/profiles_out/ (copy what is to be kept into profiles/):
#   tools/collect_r06.sh <tag>
T=${1:-a}; R=r06
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/profiles_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the bench line (REF, C3, library defaults: frames_in_flight 8) and the DDGI line
timeout 400 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${R}_${T}_bench.json 2> /dev/null
timeout 200 python $ROOT/bench.py --mode ddgi --steps 20 --warmup 5 > $OUT/${R}_${T}_ddgi_bench.json 2> /dev/null
MW=$(python -c "import json;print(json.load(open('$OUT/${R}_${T}_bench.json')).get('tuning',{}).get('march_waves',7))" 2>/dev/null || echo 7)
# kernel trace of the same command (wave split pinned to what the bench measured, so that every launch is the steady-state kernel).
# With frames in flight the launches alternate: a group's first launch traces up to eight updates, the continued update's own launch is
# empty — the AVERAGE over the launches is the time per update, which is what bench.py's roofline.kernel_ms is too.
DDGI_AQ_MARCH=$MW timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$T -o ref --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-march --no-extras > /dev/null 2>&1
python $ROOT/tools/profile_summary.py $ROOT/gpurun_out/prof_$T/ref "bench.py --steps 20 --warmup 5 (REF, frames_in_flight 8, the default) with DDGI_AQ_MARCH=$MW, the split the bench measured" > $OUT/${R}_${T}_ref_kernel_stats.txt
DDGI_FRAMES_IN_FLIGHT=1 DDGI_AQ_MARCH=$MW timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$T -o ref1 --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-march --no-extras > /dev/null 2>&1
python $ROOT/tools/profile_summary.py $ROOT/gpurun_out/prof_$T/ref1 "the same with DDGI_FRAMES_IN_FLIGHT=1: every launch traces its own update only" > $OUT/${R}_${T}_ref_fif1_kernel_stats.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$T -o ddgi --output-format csv -- python $ROOT/bench.py --mode ddgi --steps 20 --warmup 5 > /dev/null 2>&1
python $ROOT/tools/profile_summary.py $ROOT/gpurun_out/prof_$T/ddgi "bench.py --mode ddgi --steps 20 --warmup 5" > $OUT/${R}_${T}_ddgi_kernel_stats.txt
# counters (separate passes, kernel trace only); per-launch means, so one update per launch
cd $ROOT
export DDGI_AQ_MARCH=$MW DDGI_FRAMES_IN_FLIGHT=1
timeout 300 bash tools/pmc_icache.sh $T > /dev/null 2>&1; python tools/pmc_issue.py $T $OUT/${R}_pmc_${T}_issue.txt > /dev/null
timeout 400 bash tools/pmc_run.sh $T > /dev/null 2>&1; python tools/pmc_traffic.py $T $R > /dev/null
unset DDGI_AQ_MARCH DDGI_FRAMES_IN_FLIGHT
# what frames in flight is worth per slab; the N > 1 bench path with 4 ranks on this one GPU (functional)
timeout 300 python tools/fif_timing.py 2>/dev/null | grep world > $OUT/${R}_${T}_fif_timing.txt
FIF_MODE=ddgi timeout 300 python tools/fif_timing.py 2>/dev/null | grep world >> $OUT/${R}_${T}_fif_timing.txt
# where the VALU instructions go: the counters build of the headline instantiation (bursts, groups) x the marked assembly, against SQ_INSTS_VALU
DDGI_AQ_MARCH=$MW DDGI_FRAMES_IN_FLIGHT=1 timeout 200 python tools/aq_stats.py > $OUT/${R}_${T}_lane_stats.txt 2>/dev/null
python - > $OUT/${R}_${T}_valu_attribution.txt 2>&1 <<PY
import re, subprocess, sys
st = open("$OUT/${R}_${T}_lane_stats.txt").read()
m = re.search(r"counts: march_bursts (\\d+) event_groups (\\d+)", st)
bursts, groups = float(m.group(1)), float(m.group(2))
valu = float(re.search(r"SQ_INSTS_VALU\s+([0-9.e+]+)", open("$OUT/${R}_pmc_${T}_issue.txt").read()).group(1))
print(subprocess.run([sys.executable, "tools/valu_attribution.py", "--bursts", str(bursts), "--groups", str(groups), "--valu", str(valu)], capture_output=True, text=True).stdout)
print("# dynamic counts:", st.splitlines()[0])
PY
DDGI_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 $ROOT/bench.py --gpus 4 > $OUT/${R}_${T}_bench_p2p_4ranks_one_gpu.json 2> /dev/null
DDGI_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29521 $ROOT/bench.py --gpus 4 --mode ddgi > $OUT/${R}_${T}_c3_ddgi_4ranks_one_gpu.json 2> /dev/null
DDGI_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29523 $ROOT/bench.py --gpus 8 --workload c4 --steps 4 --warmup 2 > $OUT/${R}_${T}_c4_8ranks_one_gpu.json 2> /dev/null
timeout 300 python bench.py --workload c4 --steps 5 --warmup 2 > $OUT/${R}_${T}_c4_bench.json 2> /dev/null
timeout 400 python bench.py --workload c5 --mode ddgi --steps 12 > $OUT/${R}_${T}_c5_sdyn_ddgi_bench.json 2> /dev/null
# the cage samplers' kernels and counters (REF table path in 2x2x2 bricks, DDGI sampler + grouping)
bash tools/pmc_sample.sh $R $T > /dev/null 2>&1
timeout 900 python tools/pmc_sample_traffic.py $R $T > $OUT/${R}_${T}_sample_traffic.txt 2>&1
timeout 600 bash tools/pmc_blend.sh > $OUT/${R}_${T}_pmc_blend.txt 2>&1
# what HBM delivers to plain streaming kernels on this box
timeout 120 python tools/hbm_ceiling.py > $OUT/${R}_${T}_hbm_ceiling.txt 2>/dev/null
ls -la $OUT
