"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/ddgi_probe.h
declares, the wire formats have the reference's sizes, and compute refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ddgi_probe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddgi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ddgi):
    lib = C.CDLL(ddgi.library_path())
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in include/ddgi_probe.h but not exported: {missing}"
    from ddgi_amd import probe_engine

    assert sorted(probe_engine.EXPORTED_SYMBOLS) == declared


def test_wire_format_sizes(ddgi):
    # Appendix B of SURVEY.md: 48 / 32 / 48 bytes, field_origin at offset 32
    assert C.sizeof(ddgi.IrradianceField) == 48
    assert ddgi.IrradianceField.field_origin.offset == 32
    assert ddgi.IrradianceField.sqrt_rays_per_probe.offset == 20
    assert C.sizeof(ddgi.RenderSettings) == 32
    assert ddgi.RenderSettings.scene.offset == 20
    assert ddgi.PROBE_RAY_DTYPE.itemsize == 48
    assert ddgi.PROBE_RAY_DTYPE.fields["direction"][1] == 16
    assert ddgi.PROBE_RAY_DTYPE.fields["probe_info"][1] == 32


def test_abi_version_and_geometry_helpers(ddgi):
    lib = ddgi.load_library()
    assert lib.ddgi_abi_version() == 7
    f = ddgi.make_field((9, 7, 9), 11, 20)
    assert ddgi.texture_size(f) == (9 * 9 * 20, 7 * 20)  # rvpt.cpp:873-874
    assert ddgi.probe_tile_origin(f, 0) == (0, 0)
    assert ddgi.probe_tile_origin(f, 82) == (20, 20)      # probe_pass.comp:139-145
    with pytest.raises(ddgi.DDGIError):
        ddgi.probe_tile_origin(f, 9 * 7 * 9)


def test_compute_fails_loudly_without_gpu(ddgi):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ddgi.DDGIError) as ei:
        ddgi.ProbeEngine(ddgi.make_field(), ddgi.make_settings())
    assert ei.value.code == -2  # DDGI_ERR_NO_DEVICE: the product has no CPU path


def test_invalid_configurations_are_rejected(ddgi):
    lib = ddgi.load_library()
    h = C.c_void_p()
    bad = ddgi.make_field((0, 1, 1), 1, 4)
    assert lib.ddgi_create(C.byref(bad), C.byref(ddgi.make_settings()), 0, C.byref(h)) == -1
    assert b"probe_count" in lib.ddgi_last_error()
    ok = ddgi.make_field((2, 2, 3), 1, 4)
    assert lib.ddgi_create_sharded(C.byref(ok), C.byref(ddgi.make_settings()), 0, 0, 2, C.byref(h)) == -1
    assert b"divisible" in lib.ddgi_last_error()
    st = ddgi.make_settings(scene=7)
    assert lib.ddgi_create(C.byref(ok), C.byref(st), 0, C.byref(h)) == -1


def test_sample_table_layout_is_a_bijection(tmp_path):
    """The REF sampler's per-texel table stores the probes of a texel's plane in 2x2x2 bricks (csrc/ddgi_types.h): tests/box_layout_check.cpp
    walks grids with even and odd counts — one slot per probe, the inverse leads back, padding slots say so, a brick is one line, a slot is
    the sum of its axes' terms.  Host code only: built with hipcc (the header is the kernels' own), run on the CPU."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "box_layout_check"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-x", "hip", "-I" + os.path.join(root, "dynamic-diffuse-global-illumination-minecraft_amd", "csrc"),
                    os.path.join(root, "tests", "box_layout_check.cpp"), "-o", str(exe)], check=True, timeout=600)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and res.stdout.strip() == "ok", res.stdout + res.stderr
