#!/usr/bin/env python3
"""Trace time of k_probe_trace_aq against the march/event wave split, per scene (GPU box)."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd

CASES = {"c3": ((32, 16, 32), 2, 16, (1.4, 0.0, 1.0), 0), "cave": ((32, 16, 32), 4, 16, (-3.0, -1.0, -3.0), 0), "cornell": ((16, 16, 16), 1, 16, (0.0, 0.0, 15.0), 1), "house": ((32, 8, 24), 1, 16, (0.0, 0.0, 0.0), 2)}
for name, (counts, side, s, origin, scene) in CASES.items():
    row = []
    for split in ("rounds", "auto", 3, 4, 5, 6, 7, 8):
        os.environ.pop("DDGI_AQ_MARCH", None)
        if split == "rounds":
            os.environ["DDGI_TRACE_KERNEL"] = "rounds"
        elif split == "auto":
            os.environ["DDGI_TRACE_KERNEL"] = "queues"
        else:
            os.environ["DDGI_TRACE_KERNEL"] = "queues"
            os.environ["DDGI_AQ_MARCH"] = str(split)
        eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin), ddgi_amd.make_settings(scene, 8))
        eng.generate_probe_rays(seed=1)
        for _ in range(5):
            eng.probe_update()
        eng.synchronize()
        tr, _ = eng.update_history_ms(3)
        row.append("%s:%.3f" % (split, float(np.mean(tr))))
        eng.close()
    print(name, " ".join(row))
