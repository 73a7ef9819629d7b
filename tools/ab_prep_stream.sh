#!/bin/bash
# the feeler tables on their own stream (tuning "prep_stream" 2) against one preparation stream (1), GPU box: one rank's slab of an 8-way sharded
# C3 grid and the whole grid in DDGI mode at 8 frames in flight (tools/fif_timing.py), and the DDGI bench line; twice
for rep in 1 2; do for ps in 1 2; do
  echo "== prep_stream $ps"
  DDGI_PREP_STREAM=$ps FIF_MODE=ddgi FIF_WORLDS=8,1 FIF_FIFS=8 python tools/fif_timing.py 2>/dev/null | grep world
  DDGI_PREP_STREAM=$ps python bench.py --mode ddgi --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --mode ddgi: ms_per_step %.4f' % d['ms_per_step'])"
done; done
