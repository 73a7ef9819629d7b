"""Non-square ray tiles (BASELINE config C4: 512 rays per probe = a 32 x 16 tile) — host side.

The reference only knows s x s strata (rvpt.cpp:1147-1173, rvpt.h:87); ddgi_set_ray_tile / the oracle's
oracle_set_ray_tile generalise it to tile_x strata along z by tile_y strata along phi.  Here: the product's
host ray generation against the oracle's, and that a square tile is the reference's case."""
import numpy as np
import pytest


@pytest.mark.parametrize("tile", [(8, 4), (32, 16), (5, 7), (1, 9)])
def test_host_tile_rays_equal_oracle(ddgi, oracle, tile):
    counts, side, origin = (3, 2, 4), 3, (1.4, 0.0, 1.0)
    f = ddgi.make_field(counts, side, 6, origin)
    got = ddgi.generate_probe_rays_host(f, seed=1, tile=tile)
    oracle.set_ray_tile(*tile)
    want = oracle.generate_probe_rays(oracle.make_field(counts, side, 6, origin), oracle.new_rand_state(1))
    assert oracle.texture_size(oracle.make_field(counts, side, 6, origin)) == (counts[0] * counts[2] * tile[0], counts[1] * tile[1])
    oracle.set_ray_tile(0, 0)
    assert got.tobytes() == want.tobytes()
    n = tile[0] * tile[1]
    assert len(got) == 24 * n
    # ray i of a probe sits in texel (i % tile_x, i // tile_x); column <-> z stratum, row <-> phi stratum
    one = got[:n]
    assert np.array_equal(one["probe_info"][:, 1], np.arange(n) % tile[0])
    assert np.array_equal(one["probe_info"][:, 2], np.arange(n) // tile[0])
    z = one["direction"][:, 2].reshape(tile[1], tile[0])
    lo = 1.0 - 2.0 * (np.arange(tile[0]) + 1) / tile[0]
    hi = 1.0 - 2.0 * np.arange(tile[0]) / tile[0]
    assert (z >= lo - 1e-6).all() and (z <= hi + 1e-6).all()          # z = 1 - 2u, u in stratum x
    phi = np.mod(np.arctan2(one["direction"][:, 1], one["direction"][:, 0]), 2 * np.pi).reshape(tile[1], tile[0])
    row = np.arange(tile[1])[:, None]
    assert (phi >= 2 * np.pi * row / tile[1] - 1e-4).all() and (phi <= 2 * np.pi * (row + 1) / tile[1] + 1e-4).all()


def test_square_tile_is_the_reference_case(ddgi):
    f = ddgi.make_field((2, 3, 2), 4, 6, (0.0, 0.0, 15.0))
    assert ddgi.generate_probe_rays_host(f, seed=1).tobytes() == ddgi.generate_probe_rays_host(f, seed=1, tile=(6, 6)).tobytes()
    # later calls continue the rand() sequence: 2 draws per ray of the tile (Q1)
    assert ddgi.generate_probe_rays_host(f, seed=1, skip_calls=2, tile=(9, 4)).tobytes() != ddgi.generate_probe_rays_host(f, seed=1, tile=(9, 4)).tobytes()


def test_bad_tiles_are_rejected(ddgi):
    f = ddgi.make_field((2, 2, 2), 4, 6, (0.0, 0.0, 15.0))
    for tile in [(0, 4), (4, -1), (4097, 1)]:
        with pytest.raises(ddgi.DDGIError):
            ddgi.generate_probe_rays_host(f, seed=1, tile=tile)
