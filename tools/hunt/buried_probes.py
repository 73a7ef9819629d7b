#!/usr/bin/env python3
"""How many of C3's probes are BURIED — every ray black — and how deep in the rock do they stand?  (Round 6, second session: is there exact work to remove?)
For each probe: is its 16 x 16 tile all (0, 0, 0, 255) after an update, and the Chebyshev distance from its voxel to the nearest EMPTY voxel (scene 0, getBlockAt through
ddgi_scene_block_at, a region around the grid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import ndimage
import ddgi_amd
from bench import WORKLOADS
w = WORKLOADS["c3"]
field = ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"])
eng = ddgi_amd.ProbeEngine(field, ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
eng.probe_update(); eng.synchronize()
albedo = eng.read_textures()[0]           # raster, rgba8 [h, w, 4]
cx, cy, cz = w["counts"]; s = w["s"]
n_probes = cx * cy * cz
black = np.zeros(n_probes, bool)
for p in range(n_probes):
    x0, y0 = ddgi_amd.probe_tile_origin(field, p)
    t = albedo[y0:y0 + s, x0:x0 + s]
    black[p] = not t[..., :3].any()
print("probes %d, tiles entirely black (rgb 0): %d = %.1f %%" % (n_probes, black.sum(), 100.0 * black.mean()))
# occupancy around the grid
lo = np.array([-80, -60, -80]); hi = np.array([80, 60, 80])
t0 = time.time()
dim = hi - lo + 1
occ = np.zeros(dim[::-1], bool)  # [z][y][x]
lib = ddgi_amd.load_library()
for z in range(dim[2]):
    for y in range(dim[1]):
        for x in range(dim[0]):
            occ[z, y, x] = lib.ddgi_scene_block_at(0, int(lo[0] + x), int(lo[1] + y), int(lo[2] + z)) > 0
print("occupancy of %s voxels in %.1f s; solid %.1f %%" % (dim, time.time() - t0, 100.0 * occ.mean()))
dist = ndimage.distance_transform_cdt(occ, metric="chessboard")   # for solid voxels: Chebyshev distance to the nearest empty voxel
d = np.zeros(n_probes, int)
for p in range(n_probes):
    py = p // (cx * cz); rem = p - py * cx * cz; pz = rem // cx; px = rem - pz * cx
    pos = (np.array([px, py, pz]) - (np.array([cx, cy, cz]) - 1) // 2) * w["side"] + np.array(w["origin"])
    v = np.ceil(pos).astype(int) - lo
    d[p] = dist[v[2], v[1], v[0]]
for r in (0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 18, 20):
    sel = d > r
    print("distance to the nearest empty voxel > %2d: %5d probes (%.1f %%), of them black %5d; black probes NOT selected %5d" % (r, sel.sum(), 100.0 * sel.mean(), (sel & black).sum(), (black & ~sel).sum()))
print("non-black probes by distance (distance: count):", {int(k): int(((d == k) & ~black).sum()) for k in range(0, 8)})
