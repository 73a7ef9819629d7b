#!/usr/bin/env python3
"""End-to-end demo on the GPU box: probe update -> probe texture PNG -> rendered frames
(integrator_DDGI / indirect / direct) of the Cornell box and the cave.  Writes under gpurun_out/demo/."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from ddgi_amd import imageio

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "demo")
os.makedirs(out, exist_ok=True)
W, H = 640, 360
for name, counts, side, s, origin, scene, cam in [
        ("cornell", (5, 5, 5), 4, 20, (0.0, 0.0, 15.0), 1, ((0.5, 0.0, -9.0), (0, 0, 0))),
        ("cave", (9, 7, 9), 6, 20, (1.4, 0.0, 1.0), 0, ((4.0, 2.0, -14.0), (20, 5, 0)))]:
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin), ddgi_amd.make_settings(scene, 8))
    eng.generate_probe_rays(seed=1)
    eng.probe_update()
    print(name, "probe texture", imageio.dump_probe_textures(eng, os.path.join(out, name + "_probes")), eng.last_update_ms())
    camera = ddgi_amd.make_camera(*cam, fov_deg=75.0, aspect=W / H)
    for mode, tag in ((0, "ddgi"), (1, "direct"), (2, "indirect")):
        st = ddgi_amd.make_settings(scene, 8)
        st.screen_width, st.screen_height, st.render_mode = W, H, mode
        imageio.write_png(os.path.join(out, f"{name}_{tag}.png"), eng.render(camera, st))
    eng.close()
print("wrote", sorted(os.listdir(out)))
