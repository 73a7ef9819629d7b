#!/bin/bash
# blend kernels' times per build on the same GPU box:  tools/ab_blend.sh [alt names...]   (libddgi_probe_<name>.so from `make alt`)
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for rep in 1 2; do
echo "== main"; bash tools/ddgi_kernel_times.sh X=0 | grep -i "blend"
for n in "$@"; do echo "== $n"; bash tools/ddgi_kernel_times.sh DDGI_LIB=$D/libddgi_probe_$n.so | grep -i "blend"; done
done
