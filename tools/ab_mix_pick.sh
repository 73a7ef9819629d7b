#!/bin/bash
# which update's rays a mixed event group keeps (DDGI_AQ_MIX_PICK), per library: DDGI slab of an 8-way sharded C3 grid and the whole grid at 8 frames in flight; twice
D=$PWD/dynamic-diffuse-global-illumination-minecraft_amd
for rep in 1 2; do for lib in $(cd $D; ls libddgi_probe*.so | grep -v prof); do
  echo "== $lib"; DDGI_LIB=$D/$lib FIF_MODE=ddgi FIF_WORLDS=8,1 FIF_FIFS=8 python tools/fif_timing.py 2>/dev/null | grep world
done; done
