"""Strip tiling of ONE large image over several GPUs with per-octave halo exchange (BASELINE configs[4]:
a single 8192x8192 image over 8 MI355X, merged SiftData).  New functionality -- the reference has no tiling
(its arena is sized for the whole image, cuSIFT.cu:81-98); the equality target is the single-GPU result on the
whole image, which this scheme reproduces bit for bit.

Partition.  Rank k owns base rows [b_k, b_{k+1}) with b_k = k*H // P (any H: strips may differ by one row); in
octave o it owns rows [b_k >> o, b_{k+1} >> o) and a keypoint belongs to the rank that owns its integer detection
row, so nothing is found twice.  Every octave band carries HALO rows of true neighbour data above and below (none at
the real image border): 4 rows for the 9-tap blur, 1 for the extremum test and the rest for the orientation /
descriptor footprint (reach ~ 8*scale + 2 px).  A keypoint whose footprint reaches beyond the halo would sample
clamped rows instead of the neighbour's: the band kernels FLAG that on the device and the extractor raises -- loudly,
never silently different from the whole image (`strict=False` turns the error into a count).

Per octave: ScaleDown produces the OWNED rows of the next octave from the current band (it needs source rows
2r-1 .. 2r+3, inside the halo), then the ranks exchange HALO rows with their two neighbours -- one grouped
ncclSend/ncclRecv pair per neighbour through the C ABI (cusift_exchange_halos: RCCL p2p over the direct xGMI link;
<= 1.5 MiB per neighbour at 8192 wide).  Detection/description then run on each band with "clamp to the global
image, then translate" row addressing (cusift_*_band entry points), coarsest octave first like the reference.

Coarse octaves collapse onto one GPU: from the first octave in which some rank would own
fewer than HALO rows -- its halo would have to come from a rank further away than the neighbour -- every rank sends
its owned rows of that octave to rank 0 (cusift_exchange_rows), which holds the whole (small) octave image and runs
the ordinary whole-image driver on it for that octave and all coarser ones (cusift_extract_batch with that octave's
initBlur and subsampling: identical arithmetic, hence identical keypoints).  8192^2 over 8 ranks: octaves 0..4 are
tiled (1024 .. 64 owned rows per rank), octaves 5 and 6 (256 and 128 rows in total) run on rank 0.
The merged SiftData is the all-gatherv of the per-rank lists (cusift_allgatherv / cusift_amd.dist).

The rank-side driver lives behind the C ABI (csrc/sift_tiled.hip: cusift_tiled_*, mirroring the octave loop of
cuSIFT.cu:175-202) and this module is a thin caller: `StripExtractor` owns one rank's output buffers and a
capi.Tiled; `run_distributed` is cusift_tiled_extract (exchange = the C ABI communicator); `run_virtual` steps P
extractors of one process together (exchange = device copies, cusift_tiled_exchange_virtual) -- used by the
single-GPU tests and to time one rank's work.  `StripPlan` is the Python twin of the plan (tests/test_tiling_plan.py
holds it equal to cusift_tiled_plan) used by the gloo host-logic tests, which run the exchange pattern on CPU tensors
through `exchange_halos`.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi

HALO = 48  # rows of neighbour data kept above/below the owned rows of every octave band


def octave_blurs(init_blur, n_oct):
    """cuSIFT.cu:188: float totInitBlur = (float)sqrt(initBlur*initBlur + 0.5f*0.5f) / 2.0f, recursively."""
    blur = [float(init_blur)]
    for _ in range(1, n_oct):
        b = blur[-1]
        blur.append(float(np.float32(np.sqrt(b * b + 0.25)) / np.float32(2.0)))
    return blur


class StripPlan:
    """Row geometry of every (rank, octave).  Any W >= 4, any H >= world; octave sizes follow the driver's integer
    halving (cuSIFT.cu:182).  `collapse` = first octave that runs whole on rank `root` (== n_oct: none)."""

    def __init__(self, W, H, world, num_octaves, halo=HALO):
        self.W, self.H, self.world, self.halo = int(W), int(H), int(world), int(halo)
        if self.world < 1 or self.H < self.world or self.W < 1:
            raise ValueError("need world >= 1 and at least one base row per rank (W=%d H=%d world=%d)" % (W, H, world))
        if self.halo < 8:
            raise ValueError("halo must be >= 8 rows (4 blur + 1 extremum + the ScaleDown taps)")
        self.w, self.h = [self.W], [self.H]
        for _ in range(1, max(1, int(num_octaves))):
            ww, hh = self.w[-1] // 2, self.h[-1] // 2
            if ww < 1 or hh < 1:
                break
            self.w.append(ww)
            self.h.append(hh)
        self.n_oct = len(self.w)
        self.pitch = [capi.ialign_up(w, 128) for w in self.w]
        self.bounds = [(k * self.H) // self.world for k in range(self.world + 1)]
        self.root = 0
        self.collapse = self.n_oct
        for o in range(self.n_oct):
            thin = self.world > 1 and min(self.own(k, o)[1] - self.own(k, o)[0] for k in range(self.world)) < self.halo
            # the band kernels need w >= 4 and h >= 3; narrower / flatter octaves run whole as well (with one rank, too)
            if thin or self.w[o] < 4 or self.h[o] < 3:
                self.collapse = o
                break

    def own(self, rank, o):
        return self.bounds[rank] >> o, self.bounds[rank + 1] >> o

    def band(self, rank, o):
        """Global rows [lo, hi) of octave o held by `rank`: owned rows + halo (tiled octaves), owned rows only (the
        collapse octave: they are shipped to the root, no neighbour data is needed)."""
        a, b = self.own(rank, o)
        if o >= self.collapse or self.world == 1:
            return a, b
        return max(0, a - self.halo), min(self.h[o], b + self.halo)

    def tiled(self, o):
        return o < self.collapse


class StripExtractor:
    """One rank of the strip-tiled extraction (needs a GPU): output buffers + a capi.Tiled.  `comm`: a capi.Comm for
    the distributed form (run_distributed) -- the extractor then runs on the communicator's context, so kernels and
    exchanges share one stream; None for run_virtual / world 1 (the context borrows torch's current stream)."""

    def __init__(self, rank, world, W, H, params, device=None, halo=HALO, comm=None, strict=True):
        if not torch.cuda.is_available():
            raise capi.CusiftError("StripExtractor needs a GPU (no CPU fallback)")
        self.rank, self.world, self.strict = rank, world, strict
        self.params = params
        self.plan = StripPlan(W, H, world, params.num_octaves, halo)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            if comm is not None:
                self.ctx, self.owns_ctx = comm.ctx, False
            else:
                self.ctx = capi.Context(self.device.index, stream=torch.cuda.current_stream().cuda_stream)
                self.owns_ctx = True
            self.comm = comm
            # every torch-side operation on this extractor's tensors goes through the context's own stream
            h = self.ctx.stream_handle()
            self.stream = (torch.cuda.ExternalStream(h, device=self.device) if h
                           else torch.cuda.default_stream(self.device))
            self.tiled = capi.Tiled(self.ctx, comm, rank, world, W, H, params, halo)
            assert self.tiled.collapse == self.plan.collapse and self.tiled.n_oct == self.plan.n_oct
            self.max_pts = params.max_pts
            with torch.cuda.stream(self.stream):
                self.points = torch.zeros((1, self.max_pts, capi.SIFT_POINT_BYTES), dtype=torch.uint8,
                                          device=self.device)
                self.counts = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._strip = None

    def _check_strip(self, strip):
        pl = self.plan
        a, b = pl.own(self.rank, 0)
        assert strip.is_cuda and tuple(strip.shape) == (b - a, pl.W), (tuple(strip.shape), (b - a, pl.W))
        if strip.dtype != torch.float32 or strip.stride(1) != 1:
            strip = strip.to(torch.float32).contiguous()
        # the strip was produced on torch's current stream; the extractor's stream must see it complete
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._strip = strip  # kept alive until the next load
        return strip.data_ptr(), strip.stride(0)

    def load_strip(self, strip):
        """strip: this rank's owned base rows, (b_{k+1} - b_k, W) float32 on the device."""
        self.tiled.load(*self._check_strip(strip))

    def extract(self, strip):
        """The whole rank-side sequence (collective over the communicator); asynchronous."""
        ptr, pitch = self._check_strip(strip)
        self.tiled.extract(ptr, pitch, self.points.data_ptr(), self.counts.data_ptr())
        return self.points, self.counts

    def process(self):
        self.tiled.process(self.points.data_ptr(), self.counts.data_ptr())

    def check(self):
        """Blocking: raises if a keypoint's sampling footprint left the halo (its descriptor would differ from the
        whole image's; strict=False: only counts).  Returns the number of such keypoints (0 = all exact)."""
        return self.tiled.check(self.strict)

    def result(self):
        self.check()
        with torch.cuda.stream(self.stream):
            n = int(min(int(self.counts.item()), self.max_pts))
            return self.points[0, :n].cpu().numpy().view(capi.SIFT_POINT_DTYPE).reshape(-1)

    def close(self):
        self.tiled.close()
        if self.owns_ctx:
            self.ctx.close()


# ------------------------------------------------------------------------------------------------
# exchange pattern on torch tensors (the gloo host-logic twin of cusift_tiled_exchange)
# ------------------------------------------------------------------------------------------------
def exchange_halos(ext, o, group=None):
    """Tiled octave o of an object with `_views(o)` (tests/tiling_worker.py's CPU bands): HALO owned rows go to each
    neighbour, the neighbour's come in, as torch.distributed P2P ops with the layout cusift_exchange_halos uses."""
    send_up, send_dn, recv_up, recv_dn = ext._views(o)
    up, dn = _global_rank(ext.rank - 1, group), _global_rank(ext.rank + 1, group)  # P2POp peers are GLOBAL ranks
    ops = []
    if send_up is not None:
        ops.append(dist.P2POp(dist.isend, send_up, up, group))
        ops.append(dist.P2POp(dist.irecv, recv_up, up, group))
    if send_dn is not None:
        ops.append(dist.P2POp(dist.isend, send_dn, dn, group))
        ops.append(dist.P2POp(dist.irecv, recv_dn, dn, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def _global_rank(group_rank, group):
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def run_distributed(ext, strip):
    """Extract this rank's share of the tiled image (cusift_tiled_extract: every rank of the communicator calls it);
    returns (points uint8 [1,max_pts,588], counts int32 [1]).  Merge with cusift_amd.dist.SiftGatherer."""
    return ext.extract(strip)


def run_virtual(exts, strips):
    """All ranks in one process on one device: the exchanges become device-to-device copies."""
    pl = exts[0].plan
    tiles = [e.tiled for e in exts]
    for e, s in zip(exts, strips):
        e.load_strip(s)
    for o in range(min(pl.collapse + 1, pl.n_oct)):
        if o > 0:
            for e in exts:
                e.tiled.build_octave(o)
        if len(exts) > 1 or o == pl.collapse:  # one extractor: the collapse octave still moves its band into `full`
            capi.tiled_exchange_virtual(tiles, o)
    for e in exts:
        e.process()
    return [e.result() for e in exts]
