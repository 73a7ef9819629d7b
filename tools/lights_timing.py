#!/usr/bin/env python3
"""Trace time of a C3-sized grid in DDGI mode with the shipped light and with S-Dyn's four animated lights (frames 9..20), per library:
   tools/lights_timing.py [alt names...]   (libddgi_probe_<name>.so from `make alt ALTNAME=<name>`), interleaved on the same box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np
    import ddgi_amd
    from bench import WORKLOAD as w, WORKLOAD_C5 as w5
    out = []
    for four in (False, True):
        eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
        eng.set_mode(ddgi_amd.MODE_DDGI)
        if four:
            eng.set_lights(w["scene"], np.array(w5["lights"], dtype=ddgi_amd.LIGHT_DTYPE))
        eng.tune()
        st = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
        ms = []
        for f in range(20):
            st.time = 2.0 * (f + 1)
            eng.probe_update(st)
            eng.synchronize()
            if f >= 8:
                ms.append(eng.last_update_ms()["trace_ms"])
        out.append("%d light%s %.3f ms (min %.3f, march waves %d)" % (4 if four else 1, "s" if four else " ", sum(ms) / len(ms), min(ms), eng.get_tuning("march_waves")))
        eng.close()
    print(";  ".join(out))
    sys.exit(0)
D = os.path.join(ROOT, "dynamic-diffuse-global-illumination-minecraft_amd")
names = ["main"] + sys.argv[1:]
for rep in range(3):
    for n in names:
        env = dict(os.environ)
        if n != "main":
            env["DDGI_LIB"] = os.path.join(D, "libddgi_probe_%s.so" % n)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
        print("%-8s: %s" % (n, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
