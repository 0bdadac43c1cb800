"""Strip tiling of ONE large image over several GPUs with per-octave halo exchange (BASELINE configs[4]:
a single 8192x8192 image over 8 MI355X, merged SiftData).  New functionality -- the reference has no tiling
(its arena is sized for the whole image, cuSIFT.cu:81-98); the equality target is the single-GPU result on the
whole image, which this scheme reproduces bit for bit for every keypoint whose sampling footprint fits the halo.

Partition.  Rank k owns base rows [k*H/P, (k+1)*H/P); in octave o it owns rows [b_k >> o, b_{k+1} >> o) and a
keypoint belongs to the rank that owns its integer detection row, so nothing is found twice.  Every octave band
carries HALO rows of true neighbour data above and below (none at the real image border): 4 rows for the 9-tap
blur, 1 for the extremum test and the rest for the orientation/descriptor footprint (reach ~ 8*scale + 2 px).

Per octave: ScaleDown produces the OWNED rows of the next octave from the current band (it needs source rows
2r-1 .. 2r+3, inside the halo), then the ranks exchange HALO rows with their two neighbours -- one grouped
isend/irecv pair per neighbour (RCCL p2p over the direct xGMI link; <= 1.5 MiB per neighbour at 8192 wide).
Detection/description then run on each band with "clamp to the global image, then translate" row addressing
(cusift_*_band entry points), coarsest octave first like the reference.  The merged SiftData is the
all-gatherv of the per-rank lists (cusift_amd.dist).

`StripExtractor` is one rank.  `run_distributed` drives it under torch.distributed; `run_virtual` drives P
extractors in one process (exchange = device copies) -- used by the single-GPU tests and to time one rank's work.
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
    """Row geometry of every (rank, octave)."""

    def __init__(self, W, H, world, num_octaves, halo=HALO):
        self.W, self.H, self.world, self.n_oct, self.halo = int(W), int(H), int(world), int(num_octaves), int(halo)
        top = 1 << (self.n_oct - 1)
        if self.H % (self.world * top) != 0:
            raise ValueError("H=%d must be a multiple of world*2^(octaves-1)=%d" % (H, self.world * top))
        if self.W % (4 * top) != 0:
            raise ValueError("W=%d must be a multiple of 4*2^(octaves-1)=%d (float4 rows in every octave)" % (W, 4 * top))
        if (self.H // self.world) >> (self.n_oct - 1) < self.halo and self.world > 1:
            raise ValueError("strips too thin: the coarsest octave owns %d rows < halo %d"
                             % ((self.H // self.world) >> (self.n_oct - 1), self.halo))
        self.w = [self.W >> o for o in range(self.n_oct)]
        self.h = [self.H >> o for o in range(self.n_oct)]
        self.pitch = [capi.ialign_up(w, 128) for w in self.w]

    def own(self, rank, o):
        b0 = rank * (self.H // self.world)
        b1 = (rank + 1) * (self.H // self.world)
        return b0 >> o, b1 >> o

    def band(self, rank, o):
        a, b = self.own(rank, o)
        return max(0, a - self.halo), min(self.h[o], b + self.halo)


class StripExtractor:
    """One rank of the strip-tiled extraction (needs a GPU)."""

    def __init__(self, rank, world, W, H, params, device=None, halo=HALO):
        if not torch.cuda.is_available():
            raise capi.CusiftError("StripExtractor needs a GPU (no CPU fallback)")
        self.rank, self.world = rank, world
        self.params = params
        self.plan = StripPlan(W, H, world, params.num_octaves, halo)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        pl = self.plan
        with torch.cuda.device(self.device):
            self.ctx = capi.Context(self.device.index, stream=torch.cuda.current_stream().cuda_stream)
            self.bands = []
            for o in range(pl.n_oct):
                lo, hi = pl.band(rank, o)
                self.bands.append(torch.zeros((hi - lo, pl.pitch[o]), dtype=torch.float32, device=self.device))
            self.max_pts = params.max_pts
            self.points = torch.zeros((1, self.max_pts, capi.SIFT_POINT_BYTES), dtype=torch.uint8, device=self.device)
            self.counts = torch.zeros((1,), dtype=torch.int32, device=self.device)
            self.first = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self.blur = octave_blurs(params.init_blur, pl.n_oct)
        self.sub = [params.subsampling * (2.0 ** o) for o in range(pl.n_oct)]

    # ---- data movement ----
    def load_strip(self, strip):
        """strip: this rank's owned base rows, (H/P, W) float32 on the device."""
        pl = self.plan
        a, b = pl.own(self.rank, 0)
        lo, _ = pl.band(self.rank, 0)
        assert tuple(strip.shape) == (b - a, pl.W), tuple(strip.shape)
        self.bands[0][a - lo: b - lo, : pl.W] = strip
        self.counts.zero_()

    def _views(self, o):
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

    def build_octave(self, o):
        """ScaleDown the owned rows of octave o from the band of octave o-1 (cuSIFT.cu:185)."""
        pl = self.plan
        a, b = pl.own(self.rank, o)
        lo, _ = pl.band(self.rank, o)
        slo, shi = pl.band(self.rank, o - 1)
        self.ctx.scale_down_band(self.bands[o].data_ptr(), pl.pitch[o], lo, a, b, self.bands[o - 1].data_ptr(),
                                 pl.w[o - 1], shi - slo, pl.pitch[o - 1], slo, pl.h[o - 1], 0.5)

    def process_octave(self, o):
        """ExtractSiftOctave (cuSIFT.cu:204-270) on this rank's band, centres restricted to the owned rows."""
        pl, p = self.plan, self.params
        if not (p.lowest_scale < self.sub[o] * 2.0):  # cuSIFT.cu:194
            return
        a, b = pl.own(self.rank, o)
        lo, hi = pl.band(self.rank, o)
        self.first.copy_(self.counts)  # fstPts, cuSIFT.cu:243
        self.ctx.detect_band(self.bands[o].data_ptr(), pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o], a, b, self.blur[o],
                             p.peak_thresh, p.edge_thresh, self.sub[o], self.points.data_ptr(), self.max_pts,
                             self.counts.data_ptr())
        self.ctx.describe_band(self.bands[o].data_ptr(), pl.w[o], hi - lo, pl.pitch[o], lo, pl.h[o],
                               self.points.data_ptr(), self.max_pts, self.first.data_ptr(), self.counts.data_ptr(),
                               self.sub[o], p.tex_frac_bits)

    def result(self):
        n = int(min(int(self.counts.item()), self.max_pts))
        return self.points[0, :n].cpu().numpy().view(capi.SIFT_POINT_DTYPE).reshape(-1)

    def close(self):
        self.ctx.close()


def exchange_halos(ext, o, group=None):
    """One grouped isend/irecv pair per neighbour: HALO owned rows go out, the neighbour's come in."""
    send_up, send_dn, recv_up, recv_dn = ext._views(o)
    ops = []
    if send_up is not None:
        ops.append(dist.P2POp(dist.isend, send_up, ext.rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_up, ext.rank - 1, group))
    if send_dn is not None:
        ops.append(dist.P2POp(dist.isend, send_dn, ext.rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_dn, ext.rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def run_distributed(ext, strip, group=None):
    """Extract this rank's share of the tiled image; returns (points uint8 [1,max_pts,588], counts int32 [1])."""
    n = ext.plan.n_oct
    ext.load_strip(strip)
    for o in range(n):
        if o > 0:
            ext.build_octave(o)
        if ext.world > 1:
            exchange_halos(ext, o, group)
    for o in reversed(range(n)):
        ext.process_octave(o)
    return ext.points, ext.counts


def run_virtual(exts, strips):
    """All ranks in one process on one device: the halo exchange becomes device-to-device copies."""
    n = exts[0].plan.n_oct
    for e, s in zip(exts, strips):
        e.load_strip(s)
    for o in range(n):
        if o > 0:
            for e in exts:
                e.build_octave(o)
        views = [e._views(o) for e in exts]
        for k, e in enumerate(exts):
            send_up, send_dn, recv_up, recv_dn = views[k]
            if recv_up is not None:
                recv_up.copy_(views[k - 1][1])   # neighbour above sends its last owned rows down
            if recv_dn is not None:
                recv_dn.copy_(views[k + 1][0])   # neighbour below sends its first owned rows up
    for o in reversed(range(n)):
        for e in exts:
            e.process_octave(o)
    return [e.result() for e in exts]
