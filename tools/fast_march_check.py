#!/usr/bin/env python3
"""Fast (tolerance-mode) march vs the exact march on the GPU: texel differences and kernel time.
   python tools/fast_march_check.py [c2_cornell|c3_cave|cave_small ...]"""
import os
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ddgi_amd  # noqa: E402
from tests.common import CONFIGS  # noqa: E402


def run(name, fast, mode="ref", steps=8):
    counts, side, s, origin, scene = CONFIGS[name]
    with ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin), ddgi_amd.make_settings(scene, 8)) as eng:
        eng.set_tuning("fast_march", 1 if fast else 0)
        if mode == "ddgi":
            eng.set_mode(ddgi_amd.MODE_DDGI)
        else:
            eng.generate_probe_rays(seed=1)
        for _ in range(steps):
            if mode == "ddgi":
                eng.set_frame(0)
            eng.probe_update()
        eng.synchronize()
        tr, bl = eng.update_history_ms(4)
        active = eng.get_tuning("fast_march_active")
        out = eng.read_textures()[0] if mode == "ref" else eng.read_tiles()
        return out, float(np.min(tr)), active, eng.get_tuning("march_waves_measured")


if os.environ.get("FAST_CHECK_DDGI"):
    for name in sys.argv[1:] or ["cave_small", "c3_cave"]:
        (ia, da), ta, _, _ = run(name, False, "ddgi", 3)
        (ib, db), tb, active, _ = run(name, True, "ddgi", 3)
        for label, x, y in (("irradiance", ia[..., :3], ib[..., :3]), ("depth", da, db)):
            d = np.abs(x - y)
            rel = d / np.maximum(np.abs(x), 1e-3)
            print(f"{name} ddgi {label}: exact {ta:.3f} fast {tb:.3f} ms active={active}; equal {np.mean(d == 0) * 100:.3f} %  |d|<=1e-4+1e-3|x| {np.mean(d <= 1e-4 + 1e-3 * np.abs(x)) * 100:.4f} %  "
                  f"|d|<=1e-3+1e-2|x| {np.mean(d <= 1e-3 + 1e-2 * np.abs(x)) * 100:.4f} %  mean |d| {d.mean():.2e}  mean |x| {np.abs(x).mean():.3f}  max |d| {d.max():.3f}")
    sys.exit(0)
for name in sys.argv[1:] or ["c2_cornell", "c3_cave"]:
    a, ta, _, mwa = run(name, False)
    b, tb, active, mwb = run(name, True)
    d = np.abs(a[..., :3].astype(np.int32) - b[..., :3].astype(np.int32))
    n = d.size
    print(f"{name}: exact {ta:.3f} ms (march waves {mwa})  fast {tb:.3f} ms (march waves {mwb}, active={active})  speed-up {ta / tb:.3f}")
    print(f"   channels equal {np.mean(d == 0) * 100:.4f} %  within 1/255 {np.mean(d <= 1) * 100:.4f} %  mean |d| {d.mean():.5f}/255  max {d.max()}  texels differing {int((d.max(axis=-1) > 0).sum())} of {d.shape[0] * d.shape[1]}")
