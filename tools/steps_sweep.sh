#!/bin/bash
cd $GRAFT_REPO_ROOT/dynamic-diffuse-global-illumination-minecraft_amd
cp libddgi_probe.so /tmp/orig.so
for n in 12 16; do
  cp libddgi_probe_s$n.so libddgi_probe.so
  echo -n "steps/trip $n: "; cd $GRAFT_REPO_ROOT; DDGI_NO_BUILD=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; cd dynamic-diffuse-global-illumination-minecraft_amd
done
cp /tmp/orig.so libddgi_probe.so
