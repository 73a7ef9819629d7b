#!/bin/bash
# A/B of library builds on the same GPU box, interleaved, at frames_in_flight 8 and 1, REF (and DDGI with AB_DDGI=1):
#   tools/ab_fif.sh [alt names...]   (libddgi_probe_<name>.so; "main" = libddgi_probe.so)
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
NAMES=${@:-main}
one() { env $2 python bench.py --steps ${AB_STEPS:-24} --warmup 4 --no-cpu-baseline --no-extras --no-fast-march $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for i in 1 2 ${AB_REPS}; do
for n in $NAMES; do
  L="DDGI_LIB=$D/libddgi_probe_$n.so"; [ $n = main ] && L="X=0"
  echo -n "$n : ref fif8 "; one $n $L "--frames-in-flight 8"; echo -n " fif2 "; one $n $L "--frames-in-flight 2"; echo -n " fif1 "; one $n $L "--frames-in-flight 1"
  if [ -n "$AB_DDGI" ]; then echo -n " ddgi fif8 "; one $n $L "--mode ddgi --frames-in-flight 8"; echo -n " fif1 "; one $n $L "--mode ddgi --frames-in-flight 1"; fi
  echo
done; done
