"""Row geometry of the strip tiling (cusift_amd.tiling.StripPlan): pure host logic, CPU."""
import pytest

from cusift_amd.tiling import StripPlan


def test_plan_8192_over_8_ranks():
    pl = StripPlan(8192, 8192, 8, 5)
    assert pl.n_oct == 5 and pl.collapse == 5          # 1024 .. 64 owned rows per rank: all tiled
    assert pl.own(3, 0) == (3072, 4096) and pl.own(3, 4) == (192, 256)
    assert pl.band(0, 0) == (0, 1024 + 48) and pl.band(7, 4) == (448 - 48, 512)
    pl7 = StripPlan(8192, 8192, 8, 7)
    assert pl7.collapse == 5 and pl7.root == 0          # octave 5: 32 owned rows < 48-row halo -> whole on rank 0
    assert pl7.h[5] == 256 and pl7.h[6] == 128
    assert pl7.band(3, 5) == pl7.own(3, 5) == (96, 128)  # collapse octave: owned rows only, shipped to the root


@pytest.mark.parametrize("W,H,P,n", [(1000, 1531, 3, 6), (1920, 1080, 8, 5), (37, 911, 5, 9), (640, 300, 8, 4),
                                     (4096, 4097, 7, 8)])
def test_plan_partitions_every_octave(W, H, P, n):
    pl = StripPlan(W, H, P, n)
    assert pl.bounds[0] == 0 and pl.bounds[-1] == H
    sizes = [pl.bounds[k + 1] - pl.bounds[k] for k in range(P)]
    assert max(sizes) - min(sizes) <= 1
    for o in range(pl.n_oct):
        assert pl.h[o] == H >> o and pl.w[o] == W >> o
        spans = [pl.own(k, o) for k in range(P)]
        assert spans[0][0] == 0 and spans[-1][1] == pl.h[o]
        assert all(spans[k][1] == spans[k + 1][0] for k in range(P - 1))
        if pl.tiled(o) and P > 1:
            assert min(b - a for a, b in spans) >= pl.halo and pl.w[o] >= 4
            for k in range(P):
                lo, hi = pl.band(k, o)
                a, b = spans[k]
                assert lo == max(0, a - pl.halo) and hi == min(pl.h[o], b + pl.halo)
                # ScaleDown of the next octave's owned rows reads source rows 2r-1 .. 2r+3: inside this band
                if o + 1 < pl.n_oct and o + 1 <= pl.collapse:
                    na, nb = pl.own(k, o + 1)
                    if nb > na:
                        assert max(0, 2 * na - 1) >= lo and min(pl.h[o] - 1, 2 * (nb - 1) + 3) < hi
    if pl.collapse < pl.n_oct:
        oc = pl.collapse
        assert oc == 0 or pl.tiled(oc - 1)
        assert min(pl.own(k, oc)[1] - pl.own(k, oc)[0] for k in range(P)) < pl.halo or pl.w[oc] < 4


def test_plan_rejects_what_cannot_be_tiled():
    with pytest.raises(ValueError):
        StripPlan(64, 3, 4, 2)        # fewer base rows than ranks
    with pytest.raises(ValueError):
        StripPlan(64, 64, 2, 2, halo=4)
    assert StripPlan(64, 64, 1, 9).n_oct == 7  # 64 -> 1: the driver's integer halving stops at 1 px


def test_python_plan_equals_the_c_abi_plan():
    """cusift_tiled_plan (csrc/sift_tiled.hip: the plan the GPU ranks run) == StripPlan (the twin the gloo host-logic
    tests run) for every rank and octave of a grid of shapes, incl. uneven strips, collapse and narrow octaves."""
    from cusift_amd import capi

    cases = [(8192, 8192, 8, 7, 48), (8192, 8192, 8, 5, 48), (1000, 1531, 3, 6, 48), (640, 300, 8, 4, 48),
             (1024, 2048, 4, 7, 48), (512, 1024, 2, 2, 8), (64, 4096, 5, 9, 16), (4096, 3001, 7, 6, 32),
             (256, 768, 1, 3, 48), (33, 2000, 3, 8, 48), (99, 1000, 1, 6, 32), (640, 24, 1, 6, 48), (5, 40, 2, 3, 8)]
    for W, H, P, n_oct, halo in cases:
        py = StripPlan(W, H, P, n_oct, halo)
        for k in range(P):
            for o in range(py.n_oct):
                c = capi.tiled_plan(W, H, P, n_oct, halo, k, o)
                assert (c["n_octaves"], c["collapse"]) == (py.n_oct, py.collapse), (W, H, P, n_oct)
                assert (c["w"], c["h"], c["pitch"]) == (py.w[o], py.h[o], py.pitch[o])
                assert (c["own_begin"], c["own_end"]) == py.own(k, o)
                assert (c["band_begin"], c["band_end"]) == py.band(k, o), (W, H, P, k, o)
    with pytest.raises(capi.CusiftError):
        capi.tiled_plan(100, 3, 8, 4)  # fewer rows than ranks
