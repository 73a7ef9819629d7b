#!/usr/bin/env python3
"""Time of k_render_primary for a 1600x900 frame (the reference's window) of the cave over the bench grid (GPU box)."""
import os
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ddgi_amd
from bench import WORKLOAD as w

W, H = 1600, 900
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.generate_probe_rays(seed=1)
eng.probe_update()
cam = ddgi_amd.make_camera((4.0, 2.0, -14.0), (20, 5, 0), fov_deg=75.0, aspect=W / H)
img = torch.empty((H, W), dtype=torch.int32, device="cuda")
for mode, name in ((0, "DDGI (direct + probe field)"), (1, "direct"), (2, "indirect (probe field)"), (3, "colour"), (5, "depth")):
    st = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
    st.screen_width, st.screen_height, st.render_mode = W, H, mode
    for _ in range(2):
        eng.render_device(cam, st, img.data_ptr(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        eng.render_device(cam, st, img.data_ptr(), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("render_mode %d %-28s %.3f ms / frame (%.0f Mpixel/s)" % (mode, name, dt * 1e3, W * H / dt / 1e6))
