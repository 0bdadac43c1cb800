"""N>1 host path on CPU: world_size-2 (and 3) gloo jobs covering sharding and the all-gatherv of SiftData."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, mode, tmp_path, script="dist_worker.py", extra=()):
    port = free_port()
    out = str(tmp_path / ("out_" + mode))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        args = [sys.executable, os.path.join(HERE, script), out] + ([mode] if script == "dist_worker.py" else []) + \
            [str(a) for a in extra]
        procs.append(subprocess.Popen(args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, logs[rank][-3000:])
    return [np.load(out + ".rank%d.npz" % r) for r in range(world)]


def test_shard_range_partitions_the_batch():
    from cusift_amd.dist import shard_range

    for n, world in ((512, 8), (64, 1), (5, 2), (7, 3), (2, 4)):
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 3, 8) == (192, 256)  # 64 per GPU (BASELINE configs[3])


@pytest.mark.parametrize("world", [2, 3])
def test_allgatherv_ragged_counts(world, tmp_path):
    res = run_world(world, "random", tmp_path)
    max_pts = 10
    # expected: concatenation over ranks of each rank's packed valid records
    expect_chunks, expect_counts = [], []
    for r in range(world):
        pts, cnt = res[r]["local_pts"], np.clip(res[r]["local_cnt"], 0, max_pts)
        expect_chunks.append(np.concatenate([pts[i, : cnt[i]] for i in range(len(cnt))], axis=0)
                             if len(cnt) else np.zeros((0, 588), np.uint8))
        expect_counts.append(cnt)
    expect = np.concatenate(expect_chunks, axis=0)
    for r in range(world):
        for method in ("p2p", "padded"):
            np.testing.assert_array_equal(res[r][method + ".gathered"], expect)
            off = res[r][method + ".offsets"]
            assert off[0] == 0 and off[-1] == len(expect)
            for q in range(world):
                c = res[r][method + ".counts"][q]
                np.testing.assert_array_equal(c[: len(expect_counts[q])], expect_counts[q])
                assert not c[len(expect_counts[q]):].any()
                assert off[q + 1] - off[q] == expect_counts[q].sum()


def test_sharded_batch_equals_single_process(tmp_path, oracle):
    """Image-sharded extraction + all-gatherv == the single-process result, on every rank."""
    from cusift_amd import synth

    res = run_world(2, "extract", tmp_path)
    kw = dict(num_octaves=3, init_blur=0.0, peak_thresh=1.0, max_pts=512)
    single = [oracle.extract(synth.tile(1000 + i, 160, 120), **kw) for i in range(5)]
    want_counts = np.array([len(p) for p in single])
    want_bytes = np.concatenate([p.view(np.uint8).reshape(-1) for p in single])
    assert want_counts.sum() > 50
    for r in range(2):
        np.testing.assert_array_equal(res[r]["merged_counts"], want_counts)
        np.testing.assert_array_equal(res[r]["merged_bytes"], want_bytes)


@pytest.mark.parametrize("world", [2, 4])
def test_strip_tiling_halo_exchange_and_merge(world, tmp_path, oracle):
    """BASELINE configs[4] host logic on CPU: per-octave halo exchange rebuilds every band exactly (== the slice of
    the single-process pyramid), each keypoint is found by exactly one rank, and the all-gatherv merge equals the
    whole-image detection (rows within 1e-3: band-local float rows are re-based)."""
    from cusift_amd import synth
    from cusift_amd.tiling import StripPlan, octave_blurs
    from oracle_binding import SIFT_POINT_DTYPE, pitched
    from parity_utils import canonical_order

    W, H, n_oct, thresh = 256, 1536, 3, 2.0
    res = run_world(world, "tiling", tmp_path, script="tiling_worker.py", extra=(W, H, n_oct, thresh))
    plan = StripPlan(W, H, world, n_oct)
    img = synth.tile(99, W, H)
    pyr = [pitched(img)]
    for o in range(1, n_oct):
        pyr.append(oracle.scale_down(pyr[-1], plan.w[o - 1], plan.h[o - 1]))
    for r in range(world):
        for o in range(n_oct):
            lo, hi = plan.band(r, o)
            np.testing.assert_array_equal(res[r]["band%d" % o][:, : plan.w[o]], pyr[o][lo:hi, : plan.w[o]])
    blur = octave_blurs(0.0, n_oct)
    want = []
    for o in reversed(range(n_oct)):
        dog = oracle.laplace_multi(pyr[o], plan.w[o], plan.h[o], blur[o])
        c, n = oracle.find_points_multi(dog, plan.w[o], plan.h[o], thresh, 10.0, float(2 ** o), 8192)
        want.append(c[:n])
    want = canonical_order(np.concatenate(want))
    assert len(want) > 300
    for r in range(world):
        got = canonical_order(res[r]["merged"].view(SIFT_POINT_DTYPE).reshape(-1))
        assert len(got) == len(want)
        np.testing.assert_array_equal(got["coords2D"][:, 0], want["coords2D"][:, 0])
        np.testing.assert_allclose(got["coords2D"][:, 1], want["coords2D"][:, 1], atol=1e-3, rtol=0)
        np.testing.assert_array_equal(got["sharpness"], want["sharpness"])


def test_two_phase_allgatherv_with_tickets_in_flight(tmp_path):
    """begin_allgather / finish_allgather in the pipelined order of bench.py (phase 1 of the next step before phase 2
    of the current one) give, step by step, what the one-shot allgather_siftdata gives -- on every rank."""
    res = run_world(2, "pipelined", tmp_path)
    for r in res:
        for k in range(3):
            for name in ("counts", "gathered", "offsets"):
                assert np.array_equal(r["step%d.%s" % (k, name)], r["step%d.%s1" % (k, name)]), (k, name)
        assert r["step0.gathered"].shape[0] > 0
    for k in range(3):  # and both ranks hold the same gathered SiftData
        assert np.array_equal(res[0]["step%d.gathered" % k], res[1]["step%d.gathered" % k])
