#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/profiles_out/ (copy what is to be kept into profiles/):
#   tools/collect_profiles.sh <round> <tag>      e.g. r02 k
R=${1:-r02}; T=${2:-x}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/profiles_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# bench lines (REF: the headline; DDGI: trace + blend)
timeout 300 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${R}_${T}_bench.json 2> /dev/null
timeout 200 python $ROOT/bench.py --mode ddgi --steps 20 --warmup 5 > $OUT/${R}_${T}_ddgi_bench.json 2> /dev/null
# kernel traces of the same commands (the march/event split pinned to what the first update measured, so that every launch is the steady-state kernel)
MW=$(python -c "import json;print(json.load(open('$OUT/${R}_${T}_bench.json')).get('tuning',{}).get('march_waves',5))" 2>/dev/null || echo 5)
DDGI_AQ_MARCH=$MW timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$T -o ref --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-march > /dev/null 2>&1
python $ROOT/tools/profile_summary.py $ROOT/gpurun_out/prof_$T/ref "bench.py --steps 20 --warmup 5 (REF) with DDGI_AQ_MARCH=$MW, the split the first update measures for this workload" > $OUT/${R}_${T}_ref_kernel_stats.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$T -o ddgi --output-format csv -- python $ROOT/bench.py --mode ddgi --steps 20 --warmup 5 > /dev/null 2>&1
python $ROOT/tools/profile_summary.py $ROOT/gpurun_out/prof_$T/ddgi "bench.py --mode ddgi --steps 20 --warmup 5" > $OUT/${R}_${T}_ddgi_kernel_stats.txt
# counters (separate passes, kernel trace only)
cd $ROOT
export DDGI_AQ_MARCH=$MW   # (the counter passes pin the split too: to what the bench line's first update measured)
timeout 300 bash tools/pmc_icache.sh $T > /dev/null 2>&1; python tools/pmc_issue.py $T $OUT/${R}_pmc_${T}_issue.txt > /dev/null
timeout 400 bash tools/pmc_run.sh $T > /dev/null 2>&1; python tools/pmc_traffic.py $T $R > /dev/null
unset DDGI_AQ_MARCH
# slab scaling, sampler throughput, lane statistics
timeout 200 python tools/slab_timing.py > $OUT/${R}_${T}_slab_scaling.txt 2>/dev/null
timeout 200 python tools/sample_bench.py 2>/dev/null | grep mode > $OUT/${R}_${T}_sample_bench.txt
timeout 200 python tools/aq_stats.py 2>/dev/null | grep -v amdgpu > $OUT/${R}_${T}_lane_stats.txt
ls -la $OUT
# round 3 additions: the fast-march kernel's issue counters, the blend kernels' counters, the sample kernels, the WRITE_SIZE calibration
cd $ROOT
DDGI_FAST_MARCH=1 DDGI_AQ_MARCH=4 timeout 300 bash tools/pmc_icache.sh ${T}fast > /dev/null 2>&1; DDGI_FAST_MARCH=1 DDGI_AQ_MARCH=4 python tools/pmc_issue.py ${T}fast $OUT/${R}_pmc_${T}_fast_march_issue.txt > /dev/null
timeout 300 bash tools/pmc_blend.sh > $OUT/${R}_${T}_pmc_blend.txt 2>&1
timeout 400 bash tools/pmc_sample.sh $R $T > /dev/null 2>&1
timeout 200 bash tools/calib_write_size.sh $R > /dev/null 2>&1
timeout 300 python bench.py --workload c4 --steps 5 --warmup 2 > $OUT/${R}_${T}_c4_bench.json 2> /dev/null
timeout 300 python bench.py --workload c4 --mode ddgi --steps 5 --warmup 2 > $OUT/${R}_${T}_c4_ddgi_bench.json 2> /dev/null
timeout 400 python bench.py --workload c5 --mode ddgi --steps 12 > $OUT/${R}_${T}_c5_sdyn_ddgi_bench.json 2> /dev/null
ls -la $OUT
# round 3, blend: the MFMA / VALU issue microbenchmark, the depth kernel's lap timers (library built with -DDDGI_BLEND_LAPS), blend against the number of probes
timeout 120 tools/microbench/mfma_valu_coissue.bin > $OUT/${R}_mfma_valu_coissue.txt 2>&1
[ -f dynamic-diffuse-global-illumination-minecraft_amd/libddgi_probe_laps.so ] && timeout 200 python tools/blend_laps.py 2>/dev/null | grep -v amdgpu > $OUT/${R}_${T}_blend_laps.txt
timeout 400 bash tools/blend_sizes.sh > $OUT/${R}_${T}_blend_sizes.txt 2>&1
ls -la $OUT
