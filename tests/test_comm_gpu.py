"""The multi-GPU step behind the C ABI (cusift_comm_*, cusift_allgatherv_*, cusift_exchange_*: csrc/sift_comm.hip) on ONE
GPU: a world-1 communicator with self send/recv drives exactly the code an 8-rank job runs per peer -- ncclAllGather
of the counts, the device-side pack, ONE ncclGroup of ncclSend/ncclRecv -- and must reproduce the torch expression
cusift_amd.dist.pack_points.  (Two ranks cannot share a GPU under RCCL; the world > 1 host logic is covered by the
gloo tests, the hardware run by the driver's scaling bench.)"""
import numpy as np
import pytest
import torch

from cusift_amd import capi
from cusift_amd.batch import BatchExtractor
from cusift_amd.dist import SiftGatherer, make_comm, pack_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def extracted(gray1):
    imgs = np.stack([gray1, np.roll(gray1, (9, 31), axis=(0, 1)), np.full_like(gray1, 50.0), gray1[::-1].copy(),
                     np.roll(gray1, (100, 200), axis=(0, 1))])
    ex = BatchExtractor(5, 640, 480, num_octaves=3, peak_thresh=1.0, max_pts=2048)
    pts, cnt = ex.extract(ex.images_from_numpy(imgs))
    torch.cuda.synchronize()
    assert int(cnt[2]) == 0 and int(cnt[0]) > 100
    yield ex, pts, cnt
    ex.close()


@pytest.mark.parametrize("self_p2p", [True, False])
def test_allgatherv_world1_equals_pack_points(extracted, self_p2p):
    ex, pts, cnt = extracted
    want, valid = pack_points(pts, cnt, ex.max_pts)
    total = int(valid.sum())
    side = torch.cuda.Stream()
    ctx = capi.Context(0, stream=side.cuda_stream)
    comm = make_comm(ctx, self_p2p=self_p2p)
    assert comm.world == 1 and comm.rank == 0
    assert "rccl" in capi.Comm.library().lower()
    n_max = 8  # more count slots than images: padding slots must come back as zero
    g = SiftGatherer(comm, n_max, ex.max_pts, region_cap=total + 10, n_out=2, depth=2)
    for rep in range(3):  # the output ring and the tickets are reused
        # `producer`: the exchange is ordered after the extraction's stream by begin() itself -- no wait_stream here
        g.begin(pts, cnt, producer=ex.ctx)
        counts, gathered, totals = g.finish()
        side.synchronize()
        assert counts.shape == (1, n_max) and gathered.shape == (1, total + 10, 588)
        np.testing.assert_array_equal(counts[0, :5], valid.cpu().numpy())
        assert not counts[0, 5:].any()
        assert int(totals[0]) == total
        assert torch.equal(gathered[0, :total].cpu(), want.cpu()), rep
    # two exchanges in flight with DIFFERENT contents, finished oldest first: with self_p2p each ticket's packed shard
    # waits in a staging slice of its own until finish() posts it (one shared buffer sent the LATER step's records)
    cnt_b = cnt.clone()
    cnt_b[0] = cnt_b[0] // 2
    cnt_b[2] = 0
    want_b, valid_b = pack_points(pts, cnt_b, ex.max_pts)
    total_b = int(valid_b.sum())
    assert 0 < total_b < total
    torch.cuda.synchronize()
    g.begin(pts, cnt, producer=ex.ctx)
    g.begin(pts, cnt_b, producer=ex.ctx)
    with pytest.raises(capi.CusiftError, match="in flight"):
        g.begin(pts, cnt, producer=ex.ctx)
    for want_i, total_i in ((want, total), (want_b, total_b)):
        counts, gathered, totals = g.finish()
        side.synchronize()
        assert int(totals[0]) == total_i
        assert torch.equal(gathered[0, :total_i].cpu(), want_i.cpu())
    # saturated counters are clamped on the device
    cnt2 = cnt.clone()
    cnt2[1] = 10 ** 6
    torch.cuda.synchronize()
    g2 = SiftGatherer(comm, 5, ex.max_pts)
    counts, gathered, totals = g2.gather(pts, cnt2, producer=ex.ctx)
    side.synchronize()
    assert counts[0, 1] == ex.max_pts and int(totals[0]) == int(counts.sum())
    # a region that is too small is an error, not an overrun
    g3 = SiftGatherer(comm, 5, ex.max_pts, region_cap=7)
    with pytest.raises(capi.CusiftError, match="region"):
        g3.gather(pts, cnt, producer=ex.ctx)
    # finish without begin
    with pytest.raises(capi.CusiftError):
        comm.allgatherv_finish()
    # cusift_compact_gathered: regions back to back
    counts, gathered, totals = g2.gather(pts, cnt, producer=ex.ctx)
    flat = torch.zeros((total, 588), dtype=torch.uint8, device="cuda")
    capi.compact_gathered(ctx, gathered.data_ptr(), g2.region_cap, totals, flat.data_ptr(), total)
    side.synchronize()
    assert torch.equal(flat.cpu(), want.cpu())
    assert comm.host_waits() >= 0 and comm.host_wait_ms() >= 0.0
    # the trimmed wire format: the same records arrive as their 135 written floats (540 B), bit for bit
    g4 = SiftGatherer(comm, 5, ex.max_pts, region_cap=total + 3, wire_format="trimmed")
    counts, gathered, totals = g4.gather(pts, cnt, producer=ex.ctx)
    side.synchronize()
    assert gathered.shape == (1, total + 3, 540) and int(totals[0]) == total
    got = capi.expand_trimmed(gathered[0, :total].cpu().numpy().view(capi.TRIMMED_POINT_DTYPE).reshape(-1))
    ref = want.cpu().numpy().view(capi.SIFT_POINT_DTYPE).reshape(-1)
    for f in ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"):
        assert np.ascontiguousarray(got[f]).tobytes() == np.ascontiguousarray(ref[f]).tobytes(), f
    # ... and expanded on arrival (what bench.py's N > 1 step does): SiftPoint regions again, the 12 floats extraction
    # never writes zeroed, everything else bit for bit
    g5 = SiftGatherer(comm, 5, ex.max_pts, region_cap=total + 3, wire_format="trimmed", expand=True)
    counts, gathered, totals = g5.gather(pts, cnt, producer=ex.ctx)
    side.synchronize()
    assert gathered.shape == (1, total + 3, 588) and int(totals[0]) == total
    got = gathered[0, :total].cpu().numpy().view(capi.SIFT_POINT_DTYPE).reshape(-1)
    for f in ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"):
        assert np.ascontiguousarray(got[f]).tobytes() == np.ascontiguousarray(ref[f]).tobytes(), f
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos", "match_error", "empty", "coords3D"):
        assert not np.ascontiguousarray(got[f]).view(np.uint8).any(), f
    with pytest.raises(ValueError):
        SiftGatherer(comm, 5, ex.max_pts, wire_format="exact", expand=True)
    # what the LIBRARY says about the communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion)
    info = comm.info()
    assert info["lib_ranks"] in (1, -1) and info["lib_rank"] in (0, -1)
    comm.set_wire_format("exact")
    with pytest.raises(capi.CusiftError):
        comm.set_wire_format(5)
    comm.close()
    ctx.close()


def test_exchange_rows_self(ctx):
    """cusift_exchange_rows with the rank itself as the peer (self_p2p): rows are sent and received inside ONE group,
    exactly the halo step's call pattern.  Band layout [2 halo][6 own][2 halo]."""
    pitch, rows = 128, 10
    band = torch.arange(rows * pitch, dtype=torch.float32, device="cuda").reshape(rows, pitch)
    before = band.clone()
    st = torch.cuda.Stream()
    c = capi.Context(0, stream=st.cuda_stream)
    comm = make_comm(c, self_p2p=True)
    with torch.cuda.stream(st):
        st.wait_stream(torch.cuda.current_stream())
        # first 2 owned rows -> top halo, last 2 owned rows -> bottom halo (what a neighbour would receive)
        comm.exchange_rows(band.data_ptr(), pitch, rows, [(0, 2, 2, 0, 2), (0, 6, 2, 8, 2)])
    st.synchronize()
    assert torch.equal(band[0:2], before[2:4]) and torch.equal(band[8:10], before[6:8])
    assert torch.equal(band[2:8], before[2:8])
    # world 1 has no neighbours: the halo form is a no-op
    comm.exchange_halos(band.data_ptr(), pitch, 0, rows, 0, 2)
    # without self_p2p a self-addressed op is refused
    comm2 = make_comm(c)
    with pytest.raises(capi.CusiftError, match="addresses this rank"):
        comm2.exchange_rows(band.data_ptr(), pitch, rows, [(0, 2, 2, 0, 2)])
    # rows outside the band are refused on the host (RCCL would read / write them on the device)
    with pytest.raises(capi.CusiftError, match="outside the band"):
        comm.exchange_rows(band.data_ptr(), pitch, rows, [(0, 2, 2, 9, 2)])
    with pytest.raises(capi.CusiftError, match="outside the band"):
        comm.exchange_rows(band.data_ptr(), pitch, rows, [(0, 9, 2, 0, 2)])
    comm2.close()
    comm.close()
    c.close()


def test_ctx_wait_orders_two_contexts(ctx, gray1):
    """cusift_ctx_wait: a second context (its own stream) consumes what the first one produced, no host wait."""
    other = capi.Context(0)
    a = capi.DeviceBuffer(ctx, 1 << 24)
    b = capi.DeviceBuffer(other, 1 << 24)
    ctx.memset(a.ptr, 0x5A, a.nbytes)          # asynchronous on ctx's stream
    other.wait(ctx)
    capi.check(capi.lib().cusift_memcpy_d2d(other.handle, b.ptr, a.ptr, a.nbytes))
    got = b.to_numpy(np.uint8, (a.nbytes,))
    assert (got == 0x5A).all()
    other.close()


def test_tiled_refuses_a_communicator_on_another_stream(ctx):
    """Kernels and exchanges of the tiled driver must share one stream: cusift_tiled_create refuses a communicator that
    is bound to another context (a halo received on another stream would race the detection -- round 2's Python driver
    could be built that way)."""
    a, b = capi.Context(0), capi.Context(0)  # two contexts, each with a stream of its own
    comm = make_comm(a)
    prm = capi.default_params(num_octaves=3, max_pts=1024)
    with pytest.raises(capi.CusiftError, match="another context"):
        capi.Tiled(b, comm, 0, 1, 256, 256, prm)
    t = capi.Tiled(a, comm, 0, 1, 256, 256, prm)  # the communicator's own context: fine
    assert t.n_oct == 3 and t.collapse == 3
    with pytest.raises(capi.CusiftError, match="rank"):
        capi.Tiled(a, comm, 1, 2, 256, 256, prm)  # the communicator is rank 0 of 1
    t.close()
    comm.close()
    a.close()
    b.close()
