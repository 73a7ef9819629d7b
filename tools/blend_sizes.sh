#!/bin/bash
# blend kernels' times against the number of probes (GPU box): the persistent depth kernel's fixed cost and its cost per group
ROOT=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for counts in 32,8,32 32,16,32 32,32,32 64,32,32; do
  rm -rf /tmp/prof_bs
  DDGI_TIMING_COUNTS=$counts timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_bs -o t --output-format csv -- python $ROOT/tools/ddgi_timing.py > /dev/null 2>&1
  echo "== probes $counts"; python $ROOT/tools/profile_summary.py /tmp/prof_bs/t "ddgi_timing $counts" | grep -i "blend_depth\|blend_irr\|blend_mfma"
done
