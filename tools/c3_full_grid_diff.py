#!/usr/bin/env python3
"""Every texel of the C3 update against the oracle (GPU box): prints the number that differ and the first few."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ddgi_amd
from oracle import oracle_py as O
from bench import WORKLOAD as w
O.set_arith(True)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else w["seed"]
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=seed)
eng.probe_update(); a1, _ = eng.read_textures()
f = O.make_field(w["counts"], w["side"], w["s"], w["origin"])
rays = O.generate_probe_rays(f, O.new_rand_state(seed))
want, _ = O.probe_update(f, O.make_settings(w["scene"], 8), rays)
bad = (a1 != want).any(axis=-1)
print("seed", seed, "texels differing:", int(bad.sum()), "of", bad.size)
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:5]:
    print("  texel", y, x, "gpu", a1[y, x], "oracle", want[y, x])
