"""Queue kernel vs round kernel on two small scenes: same bytes? (GPU box)"""
import os
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from tests.common import CONFIGS
def run(name, env):
    for k in ("DDGI_TRACE_KERNEL",): os.environ.pop(k, None)
    os.environ.update(env)
    counts, side, s, origin, scene = CONFIGS[name]
    with ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin), ddgi_amd.make_settings(scene, 8)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update(); eng.synchronize()
        ms = eng.last_update_ms()["trace_ms"]
        return eng.read_textures()[0], ms
for name in ("c1_cornell", "cave_small"):
    a, t1 = run(name, {"DDGI_TRACE_KERNEL": "rounds"})
    b, t2 = run(name, {"DDGI_TRACE_KERNEL": "queues"})
    print(name, "equal", np.array_equal(a, b), "diff texels", int((a != b).any(axis=-1).sum()), "nonzero", int(b.any(axis=-1).sum()), "ms %.3f %.3f" % (t1, t2))
