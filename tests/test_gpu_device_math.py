"""Runs tests/device_math_check.hip on the GPU: host and device must evaluate the product's
pinned arithmetic headers to identical bits (the premise of the bit-exact parity tests)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_and_device_evaluate_pinned_arithmetic_identically(tmp_path):
    exe = tmp_path / "device_math_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-fast-math", os.path.join(HERE, "device_math_check.hip"), "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_f32_mfma_accumulates_in_k_order_like_an_fmaf_chain(tmp_path):
    """The premise of the MFMA blend kernel: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 == a k-ordered binary32 fma chain."""
    exe = tmp_path / "mfma_order_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-fast-math", os.path.join(HERE, "mfma_order_check.hip"), "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_short_reciprocal_and_square_root_sequences_are_ieee_on_their_whole_domains(tmp_path):
    """pm::sqrt_core / rcp_sqrt_core / rcp_fixed / rcp_upto_2p94 (v_sqrt_f32 + a +-1 ulp fix; v_rcp_f32 + one Newton step +
    v_div_fixup; the same behind a power-of-two pre-scale) against sqrtf(x) and `1.0f / x`: every binary32 argument of each
    function's stated domain, on the GPU, bit for bit."""
    exe = tmp_path / "exact_rcp_sqrt_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-fast-math", os.path.join(HERE, "exact_rcp_sqrt_check.hip"), "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr
