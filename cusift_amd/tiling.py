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

`StripExtractor` is one rank.  `run_distributed` drives it across processes (exchange = the C ABI communicator, or
torch.distributed for CPU tensors over gloo -- the host-logic twin used by tests); `run_virtual` drives P extractors in
one process (exchange = device copies) -- used by the single-GPU tests and to time one rank's work.
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
        if self.world > 1:
            for o in range(self.n_oct):
                thin = min(self.own(k, o)[1] - self.own(k, o)[0] for k in range(self.world)) < self.halo
                if thin or self.w[o] < 4:  # the band kernels need w >= 4; narrower octaves run whole as well
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
    """One rank of the strip-tiled extraction (needs a GPU).  `comm`: a capi.Comm bound to this extractor's context for
    the distributed form (run_distributed); None for run_virtual."""

    def __init__(self, rank, world, W, H, params, device=None, halo=HALO, comm=None, strict=True):
        if not torch.cuda.is_available():
            raise capi.CusiftError("StripExtractor needs a GPU (no CPU fallback)")
        self.rank, self.world, self.strict = rank, world, strict
        self.params = params
        self.plan = StripPlan(W, H, world, params.num_octaves, halo)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        pl = self.plan
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream()
            if comm is not None:
                self.ctx, self.owns_ctx = comm.ctx, False
            else:
                self.ctx, self.owns_ctx = capi.Context(self.device.index, stream=self.stream.cuda_stream), True
            self.comm = comm
            self.bands = []
            for o in range(min(pl.collapse + 1, pl.n_oct)):
                lo, hi = pl.band(rank, o)
                self.bands.append(torch.zeros((hi - lo, pl.pitch[o]), dtype=torch.float32, device=self.device))
            self.full = None  # the whole collapse octave, on the root
            if pl.collapse < pl.n_oct and rank == pl.root:
                oc = pl.collapse
                self.full = torch.zeros((pl.h[oc], pl.pitch[oc]), dtype=torch.float32, device=self.device)
            self.max_pts = params.max_pts
            self.points = torch.zeros((1, self.max_pts, capi.SIFT_POINT_BYTES), dtype=torch.uint8, device=self.device)
            self.counts = torch.zeros((1,), dtype=torch.int32, device=self.device)
            self.first = torch.zeros((1,), dtype=torch.int32, device=self.device)
            self.flags = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self.blur = octave_blurs(params.init_blur, pl.n_oct)
        self.sub = [params.subsampling * (2.0 ** o) for o in range(pl.n_oct)]

    # ---- data movement ----
    def load_strip(self, strip):
        """strip: this rank's owned base rows, (b_{k+1} - b_k, W) float32 on the device."""
        pl = self.plan
        a, b = pl.own(self.rank, 0)
        lo, _ = pl.band(self.rank, 0)
        assert tuple(strip.shape) == (b - a, pl.W), (tuple(strip.shape), (b - a, pl.W))
        with torch.cuda.stream(self.stream):
            self.bands[0][a - lo: b - lo, : pl.W] = strip
            self.counts.zero_()
            self.flags.zero_()

    def _views(self, o):
        """(send_up, send_dn, recv_up, recv_dn) row blocks of the band of a tiled octave (None at the image border)."""
        pl = self.plan
        a, b = pl.own(self.rank, o)
        lo, hi = pl.band(self.rank, o)
        t = self.bands[o]
        hal = pl.halo
        send_up = t[a - lo: a - lo + hal] if self.rank > 0 else None            # my first owned rows
        send_dn = t[b - lo - hal: b - lo] if self.rank < self.world - 1 else None  # my last owned rows
        recv_up = t[0: a - lo] if self.rank > 0 else None                       # halo above
        recv_dn = t[b - lo: hi - lo] if self.rank < self.world - 1 else None    # halo below
        return send_up, send_dn, recv_up, recv_dn

    def own_rows(self, o):
        """The owned rows of octave o inside this rank's band (a view)."""
        pl = self.plan
        a, b = pl.own(self.rank, o)
        lo, _ = pl.band(self.rank, o)
        return self.bands[o][a - lo: b - lo]

    def build_octave(self, o):
        """ScaleDown the owned rows of octave o from the band of octave o-1 (cuSIFT.cu:185)."""
        pl = self.plan
        a, b = pl.own(self.rank, o)
        lo, _ = pl.band(self.rank, o)
        slo, shi = pl.band(self.rank, o - 1)
        if b > a:
            self.ctx.scale_down_band(self.bands[o].data_ptr(), pl.pitch[o], lo, a, b, self.bands[o - 1].data_ptr(),
                                     pl.w[o - 1], shi - slo, pl.pitch[o - 1], slo, pl.h[o - 1], 0.5)

    def process_collapsed(self):
        """Root only: octaves >= collapse as ONE whole-image extraction of the collapse octave (its initBlur, its
        subsampling, the remaining octave count) -- the ordinary driver, so the ordinary results.  Runs first: the
        driver zeroes the counter and the coarse octaves lead the list (cuSIFT.cu:190-196)."""
        pl, p = self.plan, self.params
        oc = pl.collapse
        if oc >= pl.n_oct or self.rank != pl.root:
            return
        sub = capi.default_params(num_octaves=pl.n_oct - oc, init_blur=self.blur[oc], peak_thresh=p.peak_thresh,
                                  edge_thresh=p.edge_thresh, lowest_scale=p.lowest_scale, subsampling=self.sub[oc],
                                  max_pts=p.max_pts, tex_frac_bits=p.tex_frac_bits, fused_detect=p.fused_detect,
                                  root_sift=p.root_sift)
        self.ctx.extract_batch(self.full.data_ptr(), 1, pl.w[oc], pl.h[oc], pl.pitch[oc], pl.h[oc] * pl.pitch[oc], sub,
                               self.points.data_ptr(), self.counts.data_ptr())

    def process_octave(self, o):
        """ExtractSiftOctave (cuSIFT.cu:204-270) on this rank's band, centres restricted to the owned rows."""
        pl, p = self.plan, self.params
        if not (p.lowest_scale < self.sub[o] * 2.0):  # cuSIFT.cu:194
            return
        a, b = pl.own(self.rank, o)
        lo, hi = pl.band(self.rank, o)
        if b <= a:
            return
        with torch.cuda.stream(self.stream):
            self.first.copy_(self.counts)  # fstPts, cuSIFT.cu:243
        self.ctx.detect_band(self.bands[o].data_ptr(), pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o], a, b, self.blur[o],
                             p.peak_thresh, p.edge_thresh, self.sub[o], self.points.data_ptr(), self.max_pts,
                             self.counts.data_ptr())
        self.ctx.describe_band(self.bands[o].data_ptr(), pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o],
                               self.points.data_ptr(), self.max_pts, self.first.data_ptr(), self.counts.data_ptr(),
                               self.sub[o], p.tex_frac_bits, self.flags.data_ptr())

    def check(self):
        """Blocking: raises if a keypoint's sampling footprint left the halo (its descriptor would differ from the
        whole image's).  Returns the number of such keypoints' octave launches flagged (0 = all exact)."""
        f = int(self.flags.item())
        if f and self.strict:
            raise capi.CusiftError(
                "strip tiling: %d keypoint(s) of rank %d sample rows beyond the %d-row halo (scale too large for the "
                "halo); results would differ from the whole image -- use a larger halo or fewer ranks" %
                (f, self.rank, self.plan.halo))
        return f

    def result(self):
        self.check()
        n = int(min(int(self.counts.item()), self.max_pts))
        return self.points[0, :n].cpu().numpy().view(capi.SIFT_POINT_DTYPE).reshape(-1)

    def close(self):
        if self.owns_ctx:
            self.ctx.close()


# ------------------------------------------------------------------------------------------------
# exchange steps
# ------------------------------------------------------------------------------------------------
def exchange_halos(ext, o, group=None):
    """Tiled octave o: HALO owned rows go to each neighbour, the neighbour's come in.  GPU extractors with a
    communicator use the C ABI (one ncclGroup); anything else (CPU tensors over gloo: the host-logic twin in
    tests/tiling_worker.py) torch.distributed P2P ops with the same layout."""
    comm = getattr(ext, "comm", None)
    if comm is not None:
        pl = ext.plan
        a, b = pl.own(ext.rank, o)
        lo, hi = pl.band(ext.rank, o)
        comm.exchange_halos(ext.bands[o].data_ptr(), pl.pitch[o], a - lo, b - a, hi - b, pl.halo)
        return
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


def gather_collapse_octave(ext, group=None):
    """Every rank's owned rows of the collapse octave -> the root's whole-octave image."""
    pl = ext.plan
    oc = pl.collapse
    if oc >= pl.n_oct:
        return
    comm = getattr(ext, "comm", None)
    a, b = pl.own(ext.rank, oc)
    if ext.rank == pl.root:
        with torch.cuda.stream(ext.stream):
            ext.full[a:b].copy_(ext.own_rows(oc))
        ops = []
        for k in range(ext.world):
            ka, kb = pl.own(k, oc)
            if k != pl.root and kb > ka:
                ops.append((k, 0, 0, ka, kb - ka))
        if ops:
            comm.exchange_rows(ext.full.data_ptr(), pl.pitch[oc], ops)
    elif b > a:
        lo, _ = pl.band(ext.rank, oc)
        comm.exchange_rows(ext.bands[oc].data_ptr(), pl.pitch[oc], [(pl.root, a - lo, b - a, 0, 0)])


def run_distributed(ext, strip, group=None):
    """Extract this rank's share of the tiled image; returns (points uint8 [1,max_pts,588], counts int32 [1]).
    Merge with cusift_amd.dist.SiftGatherer (all-gatherv of SiftData)."""
    pl = ext.plan
    ext.load_strip(strip)
    for o in range(min(pl.collapse + 1, pl.n_oct)):
        if o > 0:
            ext.build_octave(o)
        if pl.tiled(o):
            if ext.world > 1:
                exchange_halos(ext, o, group)
        else:
            gather_collapse_octave(ext, group)
    ext.process_collapsed()
    for o in reversed(range(min(pl.collapse, pl.n_oct))):
        ext.process_octave(o)
    return ext.points, ext.counts


def run_virtual(exts, strips):
    """All ranks in one process on one device: the exchanges become device-to-device copies."""
    pl = exts[0].plan
    for e, s in zip(exts, strips):
        e.load_strip(s)
    for o in range(min(pl.collapse + 1, pl.n_oct)):
        if o > 0:
            for e in exts:
                e.build_octave(o)
        if pl.tiled(o):
            views = [e._views(o) for e in exts]
            for k, e in enumerate(exts):
                send_up, send_dn, recv_up, recv_dn = views[k]
                if recv_up is not None:
                    recv_up.copy_(views[k - 1][1])   # neighbour above sends its last owned rows down
                if recv_dn is not None:
                    recv_dn.copy_(views[k + 1][0])   # neighbour below sends its first owned rows up
        else:
            root = exts[pl.root]
            for k, e in enumerate(exts):
                a, b = pl.own(k, o)
                if b > a:
                    root.full[a:b].copy_(e.own_rows(o))
    for e in exts:
        e.process_collapsed()
    for o in reversed(range(min(pl.collapse, pl.n_oct))):
        for e in exts:
            e.process_octave(o)
    return [e.result() for e in exts]
