"""The ISA properties the design leans on, checked on the assembly hipcc emits for the product's flags (cross-compiled
here, no GPU): the fused detection fits two waves per SIMD without a single vector spill or scratch byte in ANY of its
instantiations (224-256 registers), the description kernel fits four (<= 128 registers, section
4.5), the HBM-bound stencils stay light, and the hot kernels contain the instruction forms they were written for."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def stencils():
    import kernel_regs

    asm = kernel_regs.assembly("sift_stencils.hip")
    return asm, {k["name"]: k for k in kernel_regs.kernels(asm)}


@pytest.fixture(scope="module")
def keypoints():
    import kernel_regs

    asm = kernel_regs.assembly("sift_keypoints.hip")
    return asm, {k["name"]: k for k in kernel_regs.kernels(asm)}


def test_detection_instantiations_fit_two_waves_without_spills(stencils):
    import kernel_regs

    asm, ks = stencils
    det = {n: k for n, k in ks.items() if "detect_fused_kernel" in n or "detect_multi_kernel" in n}
    # <ident, 588, no down> x 2, <ident, 64, no down> x 2, <ident, 64, down> x 2, the multi-octave launch
    assert len(det) == 7, sorted(det)
    for n, k in det.items():
        assert k["vgpr_count"] <= 256 and k["agpr_count"] == 0, (n, k)
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (n, k)
        assert kernel_regs.waves_per_simd(k) == 2, (n, k)
    # the next octave's image is written by the two kDown instantiations only (float2 stores)
    assert asm.count("buffer_store_dwordx2") >= 6


def test_description_kernel_fits_four_waves(keypoints):
    import kernel_regs

    asm, ks = keypoints
    k = next(v for n, v in ks.items() if "describe_all_kernel" in n)
    assert k["vgpr_count"] <= 128 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    assert kernel_regs.waves_per_simd(k) == 4
    assert k["group_segment_fixed_size"] <= 10240  # 16 one-wave workgroups per CU of 160 KB


def test_stencils_use_the_forms_they_were_written_for(stencils):
    asm, ks = stencils
    for needle, n in (("v_pk_fma_f32", 1000), ("wave_shr:1", 500), ("buffer_load_dwordx4", 20)):
        assert asm.count(needle) >= n, (needle, asm.count(needle))
    lap = next(v for n, v in ks.items() if "laplace_multi_fast_kernelILi2E" in n)
    sd = next(v for n, v in ks.items() if "scale_down_fast_kernel" in n)
    assert lap["vgpr_count"] <= 104 and lap["vgpr_spill_count"] == 0
    assert sd["vgpr_count"] <= 48 and sd["vgpr_spill_count"] == 0
