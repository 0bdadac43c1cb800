"""bench.py's gather_model: the committed PREDICTION of the multi-GPU exchange (no multi-GPU node has been reachable from
the build box).  Pure arithmetic -- checked here so that the numbers the first scaling curve is read against are right."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # definitions only: main() runs under __main__
    return mod


def test_link_rate_is_per_direction(bench):
    # one xGMI link: 153.6 GB/s bidirectional (the task statement's "7 links x ~153 GB/s") = 76.8 GB/s each way
    assert bench.XGMI_LINK_GBPS_PER_DIRECTION == pytest.approx(153.6 / 2)


def test_headline_exchange_is_link_bound(bench):
    kp, step_ms = 168816, 1.10
    m = bench.gather_model(kp, 588, step_ms, 3)
    shard = kp * 588
    assert m["bytes_per_rank_per_step"] == shard == m["bytes_per_peer_link_per_direction_per_step"]
    for W in (2, 4, 8):
        row = m["ranks"][str(W)]
        assert row["bytes_received_per_rank_per_step"] == shard * (W - 1)
        peak = row["eff_1.0"]
        assert peak["exchange_ms"] == pytest.approx(shard / 76.8e9 * 1e3, rel=1e-3)  # one shard over one link, any W
        assert peak["ms_per_step_overlapped"] == pytest.approx(max(step_ms, peak["exchange_ms"]), rel=1e-3)
        assert peak["ms_per_step_serial"] == pytest.approx(step_ms + peak["exchange_ms"], rel=1e-3)
        assert peak["weak_scaling_efficiency_overlapped"] == pytest.approx(step_ms / peak["ms_per_step_overlapped"], abs=1e-3)
        slow = row["eff_0.7"]
        assert slow["exchange_ms"] == pytest.approx(peak["exchange_ms"] / 0.7, rel=1e-3)
    assert m["ranks"]["8"]["eff_1.0"]["exchange_ms"] > step_ms and m["verdict"].startswith("link-bound")
    # the wire formats: exact trimmed 540 B and compact 160 B scale the exchange by their record size
    opts = [v for k, v in m["options"].items() if k != "unit"]
    assert sorted(round(o / m["ranks"]["8"]["eff_1.0"]["exchange_ms"], 3) for o in opts) == [round(160 / 588, 3), round(540 / 588, 3)]


def test_sparse_content_is_extraction_bound(bench):
    m = bench.gather_model(20000, 588, 1.0, 3)
    assert m["verdict"].startswith("extraction-bound")
    assert m["ranks"]["8"]["eff_1.0"]["weak_scaling_efficiency_overlapped"] == 1.0


def test_algorithmic_bytes_follow_the_survey(bench):
    # C2 per image blur 88,385,280 B, downsample 13,769,760 B, extrema 77,337,120 B
    blur, down, find = bench.algorithmic_bytes(1920, 1080, 5, 1)
    assert (blur, down, find) == (88385280, 13769760, 77337120)


def test_tiled_model_arithmetic(bench):
    """The committed prediction of BASELINE configs[4] (one 8192 x 8192 image over P ranks): bytes and the chain."""
    m = bench.tiled_model(8192, 8192, 5, 0.70, 89469)
    r8 = m["ranks"]["8"]
    assert r8["tiled_octaves"] == 5 and r8["collapse_octave"] is None and r8["exchanges_in_the_chain"] == 5
    assert r8["halo_bytes_per_neighbour_per_direction_by_octave"] == [48 * (8192 >> o) * 4 for o in range(5)]
    assert r8["merge_bytes_per_rank"] == int(89469 / 8 * 540)
    best, worst = r8["latency_20us_eff_1.0"], r8["latency_60us_eff_0.7"]
    halo_ms = sum(0.020 + b / 76.8e9 * 1e3 for b in r8["halo_bytes_per_neighbour_per_direction_by_octave"])
    assert best["halo_exchange_ms"] == pytest.approx(halo_ms, abs=2e-4)
    assert best["merge_ms"] == pytest.approx(0.020 + r8["merge_bytes_per_rank"] / 76.8e9 * 1e3, abs=2e-4)
    assert best["predicted_ms_per_image"] == pytest.approx(r8["kernel_ms_per_rank"] + best["halo_exchange_ms"] + best["merge_ms"], abs=2e-4)
    assert worst["predicted_ms_per_image"] > best["predicted_ms_per_image"]
    # latency-bound: nowhere near 8x, and the verdict says so
    assert 1.0 < best["predicted_speedup_over_one_gpu"] < 3.0 and "latency-bound" in m["verdict"]
    # more ranks than rows / halo: the thin octaves collapse onto rank 0 and cost one more exchange
    deep = bench.tiled_model(8192, 8192, 7, 0.70, 89469)["ranks"]["8"]
    assert deep["tiled_octaves"] == 5 and deep["collapse_octave"] == 5 and deep["exchanges_in_the_chain"] == 6
    assert deep["collapse_bytes_per_rank"] == (1024 >> 5) * (8192 >> 5) * 4


def test_model_plan_agrees_with_the_tiling_plan():
    """tiled_model's collapse rule is cusift_amd.tiling.StripPlan's (the Python twin of cusift_tiled_plan)."""
    import sys

    sys.path.insert(0, ROOT)
    from bench_legs.models import tiled_model
    from cusift_amd.tiling import StripPlan

    for n_oct in (5, 7, 9):
        for P in (2, 4, 8):
            plan = StripPlan(8192, 8192, P, n_oct)
            row = tiled_model(8192, 8192, n_oct, 0.7, 1000)["ranks"][str(P)]
            assert row["tiled_octaves"] == plan.collapse, (n_oct, P)
