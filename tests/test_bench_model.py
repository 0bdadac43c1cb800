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
