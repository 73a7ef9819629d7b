#!/bin/bash
# A/B of two builds on the same GPU box: libddgi_probe.so vs libddgi_probe_alt.so (make alt), interleaved
ALT=$PWD/dynamic-diffuse-global-illumination-minecraft_amd/libddgi_probe_alt.so
run() { echo -n "$1 : "; env $2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['roofline']['kernel_ms'])"; }
for i in 1 2 3; do
run main X=0
run alt DDGI_LIB=$ALT
done
