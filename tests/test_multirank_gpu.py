"""The world > 1 code of csrc/sift_comm.hip and csrc/sift_tiled.hip on ONE GPU: W ranks are W threads of this process,
each with its own context (stream) and communicator, joined by tests/fake_rccl (a test-only library with RCCL's entry
points, bound through cusift_comm_use_library; real RCCL refuses two ranks per device).  Everything the ranks run is the
product's C ABI -- per-peer offsets, the (rank +- step) % W schedule, matched halo sizes between neighbours, ragged and
empty shards, tickets in flight, the tiled driver with collapse -- only the transport underneath is the stand-in."""
import threading

import numpy as np
import pytest
import torch

from cusift_amd import capi, synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from cusift_amd.dist import SiftGatherer
from cusift_amd.tiling import StripExtractor, run_distributed, run_virtual
from fake_transport import fake_rccl_path, fake_stats
from parity_utils import canonical_order

pytestmark = pytest.mark.gpu

WORDS = 147


def run_ranks(world, fn, timeout=300):
    """fn(rank, make_comm) on `world` threads; re-raises the first exception.  make_comm(ctx) creates this rank's
    communicator on the fake transport (collective: every rank must call it)."""
    capi.comm_use_library(fake_rccl_path())  # process-global choice: every communicator created until it is reset
    errors, results = [], [None] * world
    try:
        uid = capi.comm_unique_id()

        def body(rank):
            try:
                results[rank] = fn(rank, lambda ctx, **kw: capi.Comm(ctx, uid, rank, world, **kw))
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, e))

        ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout)
        assert not any(t.is_alive() for t in ts), "a rank hung"
    finally:
        capi.comm_use_library(None)
    if errors:
        raise errors[0][1]
    return results


def synthetic_shard(rank, n_images, max_pts, seed):
    """Records whose every word identifies (rank, image, index, word), raw counters (some beyond max_pts, some zero)."""
    rng = np.random.default_rng(seed * 100 + rank)
    pts = np.zeros((max(n_images, 1), max_pts, WORDS), dtype=np.uint32)
    for i in range(n_images):
        base = (rank << 24) | (i << 16)
        pts[i] = (base + np.arange(max_pts, dtype=np.uint32)[:, None]) * 151 + np.arange(WORDS, dtype=np.uint32)[None, :]
    raw = rng.integers(0, max_pts + 1, size=max(n_images, 1)).astype(np.uint32)
    if n_images > 1:
        raw[rng.integers(0, n_images)] = 0
        raw[rng.integers(0, n_images)] = max_pts + 1000  # a saturated counter: clamped on the device
    return pts, raw


@pytest.mark.parametrize("world,fixed", [(2, False), (3, False), (8, False), (3, True)])
def test_allgatherv_multirank(world, fixed):
    max_pts, n_max = 40, 5
    images = [(r * 3 + 2) % (n_max + 1) for r in range(world)]  # ragged shard sizes
    images[world - 1] = 0  # a rank without images
    if world > 2:
        images[1] = n_max
    region_cap = n_max * max_pts
    rounds = 3
    shards = [[synthetic_shard(r, images[r], max_pts, seed) for seed in range(rounds)] for r in range(world)]
    before = fake_stats()

    def rank_fn(rank, make_comm):
        with torch.cuda.device(0):
            prod = capi.Context(0)            # the "extraction" stream
            side = capi.Context(0)            # the exchange stream
            comm = make_comm(side)
            comm.reserve(n_max, rounds, 0)
            if fixed:
                comm.set_fixed_size(True)
            d_pts = [DeviceBuffer.from_numpy(prod, shards[rank][s][0]) for s in range(rounds)]
            d_cnt = [DeviceBuffer.from_numpy(prod, shards[rank][s][1]) for s in range(rounds)]
            outs = [DeviceBuffer(side, world * region_cap * 588) for _ in range(rounds)]
            for o in outs:
                o.zero()
            side.synchronize()
            syncs_before = comm.hip_syncs()  # reserve() sized everything: nothing below may synchronise or allocate
            # all begins first (tickets in flight), then the finishes, oldest first
            for s in range(rounds):
                comm.allgatherv_begin(d_pts[s].ptr, d_cnt[s].ptr, images[rank], max_pts, n_max, outs[s].ptr, region_cap,
                                      producer=prod)
            got = []
            for s in range(rounds):
                counts, totals = comm.allgatherv_finish()
                side.synchronize()
                got.append((counts, totals, outs[s].to_numpy(np.uint32, (world, region_cap, WORDS))))
            waits = comm.host_waits()
            assert comm.hip_syncs() == syncs_before, "the exchange made a synchronising HIP call"
            comm.close()
            for b in d_pts + d_cnt + outs:
                b.free()
            side.close()
            prod.close()
            return got, waits

    res = run_ranks(world, rank_fn)
    for s in range(rounds):
        for r in range(world):
            counts, totals, regions = res[r][0][s]
            for src in range(world):
                pts, raw = shards[src][s]
                valid = np.minimum(raw[: images[src]], max_pts)
                np.testing.assert_array_equal(counts[src, : images[src]], valid)
                assert not counts[src, images[src]:].any()
                assert int(totals[src]) == int(valid.sum())
                want = np.concatenate([pts[i, : valid[i]] for i in range(images[src])]) if images[src] else \
                    np.zeros((0, WORDS), np.uint32)
                np.testing.assert_array_equal(regions[src, : len(want)], want, err_msg="round %d rank %d region %d" % (s, r, src))
    # the call log: per rank and round one counts all-gather (world sends each) and one group of world - 1 shard sends
    # (fewer where a shard is empty: an empty shard is neither sent nor received)
    after = fake_stats()
    assert after["mismatches"] == before["mismatches"] and after["timeouts"] == before["timeouts"]
    assert after["allgathers"] - before["allgathers"] == world * rounds


def test_allgatherv_compact_wire_format(ctx, gray1):
    """cusift_comm_set_wire_format(comm, 1): the ranks' records arrive as 160-byte compact records -- byte for byte what
    cusift_pack_points_compact makes of each rank's SiftData."""
    from test_compact import compact_reference

    world = 3
    prm = capi.default_params(num_octaves=3, init_blur=0.0, peak_thresh=1.0, max_pts=2048)
    imgs = [gray1, gray1[::-1].copy(), np.roll(gray1, (40, 77), axis=(0, 1))]
    region_cap = prm.max_pts

    def rank_fn(rank, make_comm):
        c = capi.Context(0)
        comm = make_comm(c)
        comm.set_wire_format(True)
        d_pts = DeviceBuffer(c, prm.max_pts * 588)
        h = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
        n = c.extract_host(imgs[rank], prm, d_pts.ptr, h)
        d_cnt = DeviceBuffer.from_numpy(c, np.array([n], np.uint32))
        out = DeviceBuffer(c, world * region_cap * 160)
        counts, totals = comm.allgatherv(d_pts.ptr, d_cnt.ptr, 1, prm.max_pts, 1, out.ptr, region_cap)
        c.synchronize()
        got = out.to_numpy(np.uint8, (world, region_cap, 160))
        comm.close()
        c.close()
        return h[:n].copy(), got, [int(t) for t in totals]

    res = run_ranks(world, rank_fn)
    for r in range(world):
        for src in range(world):
            want = compact_reference(res[src][0])
            assert res[r][2][src] == len(want) > 300
            assert res[r][1][src, : len(want)].tobytes() == want.tobytes(), (r, src)


def test_allgatherv_trimmed_wire_format(ctx, gray1):
    """cusift_comm_set_wire_format(comm, 2) at world 3: every rank's records arrive on every rank as 540-byte trimmed
    records -- the 135 floats extraction writes, bit for bit."""
    world = 3
    prm = capi.default_params(num_octaves=3, init_blur=0.0, peak_thresh=1.0, max_pts=2048)
    imgs = [gray1, gray1[::-1].copy(), np.roll(gray1, (40, 77), axis=(0, 1))]
    region_cap = prm.max_pts

    def rank_fn(rank, make_comm):
        c = capi.Context(0)
        comm = make_comm(c)
        comm.set_wire_format("trimmed")
        d_pts = DeviceBuffer(c, prm.max_pts * 588)
        h = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
        n = c.extract_host(imgs[rank], prm, d_pts.ptr, h)
        d_cnt = DeviceBuffer.from_numpy(c, np.array([n], np.uint32))
        out = DeviceBuffer(c, world * region_cap * 540)
        counts, totals = comm.allgatherv(d_pts.ptr, d_cnt.ptr, 1, prm.max_pts, 1, out.ptr, region_cap)
        # expand on arrival (cusift_expand_gathered): SiftPoint regions, one launch behind the exchange
        exact = DeviceBuffer(c, world * region_cap * 588)
        c.memset(exact.ptr, 0x5A, world * region_cap * 588)
        comm.expand_gathered(out.ptr, region_cap, totals, exact.ptr)
        c.synchronize()
        got = out.to_numpy(np.uint8, (world, region_cap, 540))
        full = exact.to_numpy(np.uint8, (world, region_cap, 588)).copy()
        info = comm.info()
        comm.close()
        c.close()
        return h[:n].copy(), got, [int(t) for t in totals], full, info

    res = run_ranks(world, rank_fn)
    for r in range(world):
        for src in range(world):
            want = res[src][0]
            assert res[r][2][src] == len(want) > 300
            got = capi.expand_trimmed(res[r][1][src, : len(want)].copy().view(capi.TRIMMED_POINT_DTYPE).reshape(-1))
            full = res[r][3][src, : len(want)].copy().view(SIFT_POINT_DTYPE).reshape(-1)
            for f in ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"):
                assert np.ascontiguousarray(got[f]).tobytes() == np.ascontiguousarray(want[f]).tobytes(), (r, src, f)
                assert np.ascontiguousarray(full[f]).tobytes() == np.ascontiguousarray(want[f]).tobytes(), (r, src, f)
            for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos", "match_error", "empty", "coords3D"):
                assert not np.ascontiguousarray(full[f]).view(np.uint8).any(), (r, src, f)
            # nothing beyond a region's records is written
            assert np.all(res[r][3][src, len(want):] == 0x5A)
        assert res[r][4]["lib_ranks"] in (world, -1)


def test_allgatherv_overflow_is_the_same_error_on_every_rank():
    world, max_pts, n_max = 3, 16, 2
    region_cap = 20  # rank 1 will hold 2 x 16 = 32 valid records

    def rank_fn(rank, make_comm):
        ctx = capi.Context(0)
        comm = make_comm(ctx)
        pts = np.zeros((n_max, max_pts, WORDS), np.uint32)
        cnt = np.array([16, 16] if rank == 1 else [3, 1], np.uint32)
        d_p, d_c = DeviceBuffer.from_numpy(ctx, pts), DeviceBuffer.from_numpy(ctx, cnt)
        out = DeviceBuffer(ctx, world * region_cap * 588)
        with pytest.raises(capi.CusiftError, match="region"):
            comm.allgatherv(d_p.ptr, d_c.ptr, n_max, max_pts, n_max, out.ptr, region_cap)
        # the communicator is still usable (nothing was left half posted)
        cnt2 = np.array([2, 2], np.uint32)
        ctx.h2d(d_c.ptr, cnt2)
        counts, totals = comm.allgatherv(d_p.ptr, d_c.ptr, n_max, max_pts, n_max, out.ptr, region_cap)
        ctx.synchronize()
        assert list(totals) == [4] * world
        comm.close()
        ctx.close()

    run_ranks(world, rank_fn)


def test_exchange_halos_four_ranks_and_size_mismatch():
    world, pitch, halo, own = 4, 128, 3, 10

    def band_of(rank):
        top = halo if rank > 0 else 0
        bot = halo if rank < world - 1 else 0
        rows = top + own + bot
        b = np.full((rows, pitch), -1.0, np.float32)
        b[top: top + own] = (1000 * rank + np.arange(own, dtype=np.float32))[:, None] + np.arange(pitch, dtype=np.float32)[None, :] / 1024
        return b, top, bot

    def rank_fn(rank, make_comm):
        ctx = capi.Context(0)
        comm = make_comm(ctx)
        b, top, bot = band_of(rank)
        d = DeviceBuffer.from_numpy(ctx, b)
        comm.exchange_halos(d.ptr, pitch, top, own, bot, halo)
        got = d.to_numpy(np.float32, b.shape)
        # a neighbour pair that disagrees about the halo depth must be an error on both sides, not a hang or an overrun
        bad = None
        if world > 1:
            try:
                comm.exchange_halos(d.ptr, pitch, top, own, bot, halo if rank != 1 else halo - 1)
            except capi.CusiftError as e:
                bad = str(e)
        comm.close()
        ctx.close()
        return got, bad

    before = fake_stats()
    res = run_ranks(world, rank_fn)
    for r in range(world):
        got, bad = res[r]
        b, top, bot = band_of(r)
        np.testing.assert_array_equal(got[top: top + own], b[top: top + own])
        if r > 0:
            up, utop, _ = band_of(r - 1)
            np.testing.assert_array_equal(got[:top], up[utop + own - halo: utop + own])
        if r < world - 1:
            dn, dtop, _ = band_of(r + 1)
            np.testing.assert_array_equal(got[top + own:], dn[dtop: dtop + halo])
        # rank 1 sent 2 rows where ranks 0 and 2 expected 3: those three see the mismatch; rank 3 does not
        assert (bad is not None) == (r in (0, 1, 2)), (r, bad)
    assert fake_stats()["mismatches"] > before["mismatches"]


@pytest.mark.parametrize("W,H,P,n_oct,blur,thresh,collapse", [
    (1024, 2048, 4, 4, 1.0, 3.0, 4),   # every octave tiled
    (1024, 2048, 4, 7, 1.0, 3.0, 4),   # octaves 4..6 collapse onto rank 0 (gather of rows through the transport)
    (1000, 1531, 3, 6, 0.5, 2.0, 4),   # uneven strips, ragged widths
])
def test_tiled_distributed_equals_virtual_and_whole(ctx, W, H, P, n_oct, blur, thresh, collapse):
    """StripExtractor's distributed form (cusift_tiled_extract over a communicator, one thread per rank) == run_virtual
    == the whole image, bit for bit; then the merge: every rank's all-gatherv holds the whole image's SiftData."""
    img = synth.tile(77, W, H, preblur=blur)
    prm = capi.default_params(num_octaves=n_oct, init_blur=blur, peak_thresh=thresh, max_pts=65536)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    want = canonical_order(h_pts[:n])
    d_pts.free()
    dev = torch.device("cuda", 0)
    full = torch.from_numpy(img).to(dev)
    torch.cuda.synchronize()

    exts = [StripExtractor(k, P, W, H, prm, device=dev) for k in range(P)]
    assert exts[0].plan.collapse == collapse
    bounds = exts[0].plan.bounds
    virt = run_virtual(exts, [full[bounds[k]:bounds[k + 1]] for k in range(P)])
    for e in exts:
        e.close()
    region_cap = 32768

    def rank_fn(rank, make_comm):
        with torch.cuda.device(dev):
            c = capi.Context(0)  # a context with a stream of its OWN: not torch's current stream
            comm = make_comm(c)
            ext = StripExtractor(rank, P, W, H, prm, device=dev, comm=comm)
            pts, cnt = run_distributed(ext, full[bounds[rank]:bounds[rank + 1]])
            mine = ext.result()
            g = SiftGatherer(comm, 1, prm.max_pts, region_cap=region_cap, device=dev)
            counts, gathered, totals = g.gather(pts, cnt)
            c.synchronize()
            merged = np.concatenate([x.cpu().numpy() for x in SiftGatherer.regions(gathered, totals)])
            ext.close()
            comm.close()
            c.close()
            return mine, merged.view(SIFT_POINT_DTYPE).reshape(-1), counts

    res = run_ranks(P, rank_fn)
    for k in range(P):
        a, b = canonical_order(res[k][0]), canonical_order(virt[k])
        assert a.tobytes() == b.tobytes() or _same_extracted(a, b), "rank %d differs from its virtual twin" % k
    for k in range(P):
        merged = canonical_order(res[k][1])
        assert len(merged) == len(want)
        assert _same_extracted(merged, want), "rank %d's merged SiftData differs from the whole image" % k
        np.testing.assert_array_equal(res[k][2][:, 0], [len(v) for v in virt])


def _same_extracted(a, b):
    """The fields extraction writes (the others are left as they were -- uninitialised in the reference)."""
    if len(a) != len(b):
        return False
    return all(np.array_equal(a[f], b[f], equal_nan=True)
               for f in ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"))
