import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ctypes as C
import ddgi_amd
from oracle import oracle_py as O
O.set_arith(True)
def check(counts, side, s, origin, scene, lights, nprobe=8, frames=1, seed=3):
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin), ddgi_amd.make_settings(scene, 8))
    eng.set_mode(ddgi_amd.MODE_DDGI)
    if lights is not None: eng.set_lights(scene, lights)
    for fr in range(frames): eng.probe_update(ddgi_amd.make_settings(scene, 8, time=2.0*fr))
    irr, dep = eng.read_tiles(); eng.close()
    f = O.make_field(counts, side, s, origin)
    P = counts[0]*counts[1]*counts[2]
    rng = np.random.default_rng(seed); bad = []
    for p in rng.choice(P, size=nprobe, replace=False):
        ib = np.zeros(256, np.float32); db = np.zeros(512, np.float32)
        for fr in range(frames):
            O.lib().oracle_ddgi_update(C.byref(f), C.byref(O.make_settings(scene, 8, time=2.0*fr)),
                None if lights is None else np.ascontiguousarray(lights).ctypes.data_as(C.c_void_p), 0 if lights is None else len(lights), C.c_uint32(fr),
                C.c_void_p(ib.ctypes.data - int(p)*1024), C.c_void_p(db.ctypes.data - int(p)*2048), None, int(p), 1, 1)
        d = np.abs(irr[p].reshape(-1) - ib).max(); dd = np.abs(dep[p].reshape(-1) - db).max()
        if d > 0 or dd > 0: bad.append((int(p), float(d), float(dd)))
    print(counts, side, "lights", None if lights is None else len(lights), "frames", frames, "bad", bad)
L4 = np.array([(20.0,(1,1,1),(4,17.5,8.5)),(10.0,(1,.5,.1),(0,2,0)),(10.0,(.1,1.1,1),(5,0,0)),(10.0,(1.1,0,1.1),(0,5,0))], dtype=ddgi_amd.LIGHT_DTYPE)
check((32,16,32), 2, 16, (1.4,0,1), 0, None)
check((32,16,32), 2, 16, (1.4,0,1), 0, L4)
check((64,32,64), 1, 16, (1.4,0,1), 0, None)
check((128,64,128), 1, 16, (1.4,0,1), 0, None)
check((128,64,128), 1, 16, (1.4,0,1), 0, L4)
check((128,64,128), 1, 16, (1.4,0,1), 0, L4, frames=2)
