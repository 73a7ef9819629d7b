"""live reconfiguration of the probe grid (rvpt.cpp:661-755) with carry-over of
the tiles of probes whose world position survives."""
import numpy as np
import pytest


def _positions(counts, side, origin):
    """World position per axis, in the reference's float arithmetic (rvpt.cpp:1199-1205)."""
    return [(np.arange(c, dtype=np.int32) - (c - 1) // 2).astype(np.float32) * np.float32(side) + np.float32(o) for c, o in zip(counts, origin)]


def _carry_map(new, old):
    """new/old = (counts, side, origin) -> int array [P_new] of old reference probe indices (p = y*cx*cz + z*cx + x) or -1."""
    pn, po = _positions(*new), _positions(*old)
    (cx, cy, cz), (ox, oy, oz) = new[0], old[0]
    m = np.full(cx * cy * cz, -1, dtype=np.int64)
    for y in range(cy):
        for z in range(cz):
            for x in range(cx):
                j = [np.nonzero(po[a] == pn[a][i])[0] for a, i in enumerate((x, y, z))]
                if all(len(k) for k in j):
                    m[y * cx * cz + z * cx + x] = j[1][0] * ox * oz + j[2][0] * ox + j[0][0]
    return m


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("new", [
    ((4, 3, 4), 6, (6.0, 0.0, 15.0)),     # the volume scrolls by one cell in x
    ((6, 3, 2), 6, (0.0, 0.0, 15.0)),     # more probes in x, fewer in z (the centre index moves)
    ((4, 3, 4), 3, (0.0, 0.0, 15.0)),     # half the spacing: every second probe coincides with an old one
    ((4, 3, 4), 6, (0.5, 0.0, 15.0)),     # a shift that is no multiple of the spacing: nothing survives
])
def test_ddgi_reconfigure_carries_surviving_probes(ddgi, oracle, new):
    old = ((4, 3, 4), 6, (0.0, 0.0, 15.0))
    s, scene, hyst = 4, 1, 0.8
    with ddgi.ProbeEngine(ddgi.make_field(old[0], old[1], s, old[2], hysteresis=hyst), ddgi.make_settings(scene, 4)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        for _ in range(2):
            eng.probe_update()
        irr0, dep0 = eng.read_tiles()
        eng.reconfigure(ddgi.make_field(new[0], new[1], s, new[2], hysteresis=hyst), carry_over=True)
        irr1, dep1 = eng.read_tiles()
        eng.probe_update()                       # frame index 2: the sequence goes on
        irr2, dep2 = eng.read_tiles()
    m = _carry_map(new, old)
    want_irr = np.where(m[:, None, None, None] >= 0, irr0[np.maximum(m, 0)], 0).astype(np.float32)
    want_dep = np.where(m[:, None, None, None] >= 0, dep0[np.maximum(m, 0)], 0).astype(np.float32)
    assert np.array_equal(_bits(irr1), _bits(want_irr)) and np.array_equal(_bits(dep1), _bits(want_dep))
    if new[2][0] == 0.5:
        assert (m < 0).all()
    else:
        assert (m >= 0).any() and irr1[m >= 0].any()
    # the next update blends into the carried tiles exactly as the oracle does
    f = oracle.make_field(new[0], new[1], s, new[2], hysteresis=hyst)
    oracle.ddgi_update(f, oracle.make_settings(scene, 4), 2, want_irr, want_dep)
    assert np.array_equal(_bits(irr2), _bits(want_irr)) and np.array_equal(_bits(dep2), _bits(want_dep))


@pytest.mark.gpu
def test_ref_reconfigure_carry_and_clear(ddgi):
    old = ((3, 2, 3), 8, (0.0, 0.0, 15.0))
    with ddgi.ProbeEngine(ddgi.make_field(old[0], old[1], 8, old[2]), ddgi.make_settings(1, 4)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        a0, _ = eng.read_textures()
        eng.reconfigure(ddgi.make_field((3, 2, 3), 8, 8, (8.0, 0.0, 15.0)), carry_over=True)   # scroll by one cell
        a1, _ = eng.read_textures()
        with pytest.raises(ddgi.DDGIError):
            eng.probe_update()                   # rays must be regenerated, as after recreate_probe_textures
        eng.reconfigure(ddgi.make_field((3, 2, 3), 8, 4, (8.0, 0.0, 15.0)), carry_over=True)   # other tile size: cleared
        a2, _ = eng.read_textures()
    # reference raster: tile column = z*cx + x, tile row = y; tiles are 8x8 texels
    t0 = a0.reshape(2, 8, 9, 8, 4)   # [y, ty, z*3+x, tx, c]
    t1 = a1.reshape(2, 8, 9, 8, 4)
    for z in range(3):
        for x in range(3):
            if x + 1 < 3:
                assert np.array_equal(t1[:, :, z * 3 + x], t0[:, :, z * 3 + x + 1])
            else:
                assert not t1[:, :, z * 3 + x].any()
    assert a0.any() and not a2.any()


@pytest.mark.gpu
@pytest.mark.parametrize("s_old,s_new", [(32, 16), (16, 32)])
def test_ddgi_reconfigure_to_another_ray_count_keeps_the_weight_tiles_right(ddgi, oracle, s_old, s_new):
    """Round 5's advisor finding: the two DDGI blend-weight buffers (frame f's tiles in buffer f & 1; the NEXT frame's are made on the
    preparation stream beside this frame's blend) were addressed with the current ray count's size by ddgi_probe_update and with the
    allocation's — which never shrinks — by the preparation: after a reconfiguration to FEWER rays every odd frame blended from stale
    weights.  1024 -> 256 rays (and back up), six frames each, against the oracle frame by frame."""
    counts, side, origin, scene, hyst = (4, 3, 4), 8, (1.4, 0.0, 1.0), 0, 0.8

    def field(mod, s):
        return mod.make_field(counts, side, s, origin, hysteresis=hyst)

    want_irr, want_dep = oracle.new_tiles(field(oracle, s_old))
    with ddgi.ProbeEngine(field(ddgi, s_old), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        frame = 0
        for s in (s_old, s_new):
            if s != s_old:
                eng.reconfigure(field(ddgi, s), carry_over=True)   # (same probes: every tile is carried)
            for _ in range(6):
                st = dict(time=2.0 * (frame + 1))
                eng.probe_update(ddgi.make_settings(scene, 8, **st))   # back to back: the next frame's weights are made ahead
                oracle.ddgi_update(field(oracle, s), oracle.make_settings(scene, 8, **st), frame, want_irr, want_dep)
                frame += 1
            irr, dep = eng.read_tiles()
            assert np.array_equal(_bits(irr), _bits(want_irr)), f"irradiance tiles differ after the frames at {s * s} rays"
            assert np.array_equal(_bits(dep), _bits(want_dep)), f"depth tiles differ after the frames at {s * s} rays"
