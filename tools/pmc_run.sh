#!/bin/bash
# Collects rocprofv3 PMC passes for the bench workload on the GPU box (separate passes; no
# sys/hip/hsa trace domains together with --pmc).  Usage: tools/pmc_run.sh <tag>
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DDGI_AQ_MARCH=${DDGI_AQ_MARCH:-5}   # every launch is the steady-state kernel (no wave-split measurement launches among them)
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-march --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $OUT -o sq1 --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $OUT -o sq2 --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT -o grbm --output-format csv -- $BENCH > $OUT/grbm.log 2>&1
ls -R $OUT | head -40
