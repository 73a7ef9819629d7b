#!/bin/bash
# A/B of builds on the same GPU box, interleaved:  tools/ab.sh [alt names...]   (libddgi_probe_<name>.so from `make alt ALTNAME=<name>`; default: alt)
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
NAMES=${@:-alt}
run() { echo -n "$1 : "; env $2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d.get('fast_march',{}); print('%.3f ms  (march waves %s)   fast %.3f ms' % (d['roofline']['kernel_ms'], d['tuning']['march_waves'], f.get('kernel_ms', float('nan'))))"; }
for i in 1 2 3; do
run main X=0
for n in $NAMES; do run $n DDGI_LIB=$D/libddgi_probe_$n.so; done
done
