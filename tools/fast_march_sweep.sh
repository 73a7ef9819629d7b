#!/bin/bash
# A/B of fast-march builds on one box: tools/fast_march_sweep.sh <alt names...>   (libddgi_probe_<name>.so from `make alt`)
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for rep in 1 2; do
  echo -n "main: "; python tools/fast_march_check.py c3_cave 2>/dev/null | head -1
  for n in "$@"; do echo -n "$n: "; DDGI_LIB=$D/libddgi_probe_$n.so python tools/fast_march_check.py c3_cave 2>/dev/null | head -1; done
done
