#!/bin/bash
# DDGI blend on a rank's slab (GPU box): one launch for depth + irradiance, or two — by the tuning "blend_merge" (half depth groups per CU)
for m in 0 1 2 4; do echo "== DDGI_BLEND_MERGE=$m"; DDGI_BLEND_MERGE=$m python tools/slab_timing.py 2>/dev/null | grep " ddgi "; done
