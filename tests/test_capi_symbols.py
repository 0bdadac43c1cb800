"""The C-ABI library loads on a CPU-only machine and exports every symbol the four headers include/cusift_amd*.h declare."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("cusift_amd.h", "cusift_amd_stages.h", "cusift_amd_multigpu.h", "cusift_amd_extras.h")


def declared_symbols(headers=HEADERS):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(cusift_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_the_front_door_is_small():
    """include/cusift_amd.h is the drop-in boundary: what include/cuSIFT.h is built on, the batch driver, the pipeline --
    not the stage entry points, not the multi-GPU half (round-4 verdict: 103 functions in one header for a reference
    surface of ~15).  cusift_amd_all.h is the four headers together."""
    core = declared_symbols(("cusift_amd.h",))
    assert 30 <= len(core) <= 50, len(core)
    for n in ("cusift_extract", "cusift_extract_host", "cusift_extract_batch", "cusift_scale_down", "cusift_rootsift",
              "cusift_pipe_submit", "cusift_ctx_create", "cusift_malloc", "cusift_event_record"):
        assert n in core, n
    for n in core:
        assert not n.startswith(("cusift_comm_", "cusift_tiled_", "cusift_allgatherv", "cusift_laplace", "cusift_match")), n
    allh = open(os.path.join(ROOT, "include", "cusift_amd_all.h")).read()
    assert all(h in allh for h in HEADERS)


def test_library_exports_every_declared_symbol():
    from cusift_amd import capi

    names = declared_symbols()
    assert len(names) >= 30
    handle = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    # and the binding covers exactly the declared surface
    assert sorted(capi.SIGNATURES) == names


def test_binding_loads_and_reports_version():
    from cusift_amd import capi

    lib = capi.lib()
    assert b"cusift_amd" in lib.cusift_version()
    p = capi.default_params()
    assert p.num_octaves == 5 and p.edge_thresh == 10.0 and p.tex_frac_bits == 8 and p.max_pts == 1024
    assert p.fused_detect == 1 and p.root_sift == 0 and p.concurrent_batches == 1  # the struct's tail: layout check


def test_point_record_is_588_bytes():
    from cusift_amd import capi

    assert capi.SIFT_POINT_DTYPE.itemsize == 588
    assert capi.SIFT_POINT_DTYPE.fields["data"][1] == 64


def test_laplace_taps_host_table_matches_oracle(oracle):
    """Host-only entry point: the tap table the product uploads equals the oracle's, bit for bit."""
    from cusift_amd import capi

    for blur in (0.0, 0.5, 0.7, 1.0, 0.55901699):
        np.testing.assert_array_equal(capi.laplace_taps(blur), oracle.laplace_taps(blur))


def test_no_device_is_an_error_not_a_fallback():
    from cusift_amd import capi

    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.CusiftError):
        capi.Context(0)


def test_product_does_not_reference_the_oracle():
    """The shipped package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "cusift_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libsift_oracle" not in text and "oracle_binding" not in text, os.path.join(dirpath, f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", f), errors="ignore").read()
        assert "sift_oracle" not in text, f


def test_canonical_sort_is_a_host_function(oracle, gray1):
    """cusift_sort_points_host needs no GPU: octave coarsest first, then y, x, scale -- the order the parity tests'
    canonical_order() uses; a shuffled copy sorts back to the same bytes."""
    from cusift_amd import capi
    from parity_utils import canonical_order

    pts = oracle.extract(gray1, num_octaves=4, peak_thresh=1.0, max_pts=4096).view(capi.SIFT_POINT_DTYPE).copy()
    assert len(pts) > 500
    rng = np.random.default_rng(1)
    a, b = pts[rng.permutation(len(pts))].copy(), pts[rng.permutation(len(pts))].copy()
    capi.sort_points(a)
    capi.sort_points(b)
    assert a.tobytes() == b.tobytes()
    ref = canonical_order(pts)
    for f in ("subsampling", "coords2D", "scale"):
        np.testing.assert_array_equal(a[f], ref[f])
