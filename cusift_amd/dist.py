"""Multi-GPU layer: one process per GPU, images sharded across ranks, all-gatherv of SiftData.

The reference is single-GPU (no collective call sites); BASELINE configs[3] adds the
sharded batch.  Images are independent, so the data path has NO collective; the only exchange is the
final all-gatherv of the variable-length SiftPoint lists:

  1. all_gather of the per-image counts                      (n_local int32 per rank, tiny)
  2. exact-count exchange of the packed 588-byte records, rank to rank.  xGMI is point-to-point
     (7 links per GPU): one send/recv pair per peer in a single group (RCCL grouped p2p), so every shard
     travels over its own link instead of hopping round a ring.

On GPUs the exchange is the C ABI's (cusift_comm_* / cusift_allgatherv_*, csrc/sift_comm.hip: RCCL called directly
from C++) and this module is a thin caller: `make_comm` hands the communicator's unique id to every rank through
torch.distributed, `SiftGatherer` owns the preallocated output buffers.  The torch.distributed implementation below
(`begin_allgather` / `finish_allgather` / `allgather_siftdata`) is the host-logic twin used with CPU tensors over
`gloo` (tests/test_dist_gloo.py); it packs the ranks' records back to back (the layout cusift_compact_gathered makes of
the C ABI's fixed regions).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .capi import SIFT_POINT_BYTES


def make_comm(ctx, group=None, self_p2p=False):
    """A capi.Comm (RCCL communicator behind the C ABI) over the ranks of `group`, bound to `ctx` (its device and
    stream).  Rank 0 creates the unique id; torch.distributed only carries those 128 bytes."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ids = [capi.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=_global_rank(0, group), group=group)
    else:
        rank, world, ids = 0, 1, [capi.comm_unique_id()]
    return capi.Comm(ctx, ids[0], rank, world, self_p2p=self_p2p)


class SiftGatherer:
    """All-gatherv of SiftData on GPUs through the C ABI, with everything allocated once.

        g = SiftGatherer(comm, n_images_max=64, max_pts=32768, region_cap=64 * 8192, depth=4)
        g.begin(points, counts, producer=ex.ctx)   # asynchronous: ordered after the extraction on ex.ctx's stream;
                                                   # counts exchange + the local shard packed into its region
        ex.ctx.wait(comm.ctx)                      # (before the NEXT extraction into `points`: the pack reads them)
        ...                                        # enqueue further extractions meanwhile -- up to `depth` begins may
                                                   # be outstanding
        counts, gathered, totals = g.finish()      # of the OLDEST begin: host read of the counts (arrived long ago in a
                                                   # pipelined loop -> no wait), then the grouped ncclSend/ncclRecv
    `gathered` is a [world, region_cap, 588] uint8 view of an internal ring buffer (n_out deep, n_out >= depth), valid
    until n_out begins later; rank r's records are gathered[r, :totals[r]] in image order (`regions()` lists them).
    `region_cap` = records one rank's region holds (default: the worst case n_images_max * max_pts).
    `compact=True`: the records travel and arrive as 160-byte cusift_compact_point (capi.COMPACT_POINT_DTYPE; exact
    header fields, 8-bit descriptor) -- `gathered` is then [world, region_cap, 160].  `wire_format="trimmed"`: as 540-byte
    cusift_trimmed_point (the 135 floats extraction writes: exact; capi.TRIMMED_POINT_DTYPE, capi.expand_trimmed).
    `wire_format="trimmed", expand=True` (what bench.py uses for N > 1): the records TRAVEL as trimmed records and are
    expanded on arrival (cusift_expand_gathered, one launch behind the exchange), so finish() returns [world, region_cap,
    588] SiftPoint regions exactly as the exact format does -- every rank ends the step holding SiftData of all images --
    for 8 % fewer bytes per xGMI link (the 12 floats dropped are never written by extraction and uninitialised in the
    reference, cuSIFT.cu:24,29; they arrive as zeros).
    `expand=True` is an EXTRACTION-ONLY format: score, ambiguity, match, match_xpos, match_ypos, match_error, empty and
    coords3D are not carried -- match results written into the records before the gather (MatchSiftData) arrive zeroed,
    and the gathered SiftData is byte-identical to the producer's only in the 135 floats extraction writes.  Gather records
    that carry match results with wire_format="exact" (bench.py --gather-exact, the format to compare metrics of rounds
    1-4 with).  It also keeps a second ring of exact-size output buffers (world * region_cap * 588 bytes * n_out) beside
    the trimmed arrival buffers.
    The records and counters handed to begin() must stay untouched until the pack enqueued by begin() has run: begin()
    returns an event recorded behind it -- `producer_stream.wait_event(ev)` (or `producer_ctx.wait(comm.ctx)`) orders
    the producer's next write after the pack without a host wait.  The tensors themselves are kept alive by this object
    until that event has fired."""

    def __init__(self, comm, n_images_max, max_pts, region_cap=None, device=None, n_out=2, depth=1, fixed_size=False,
                 compact=False, wire_format=None, expand=False):
        self.comm, self.n_max, self.max_pts = comm, int(n_images_max), int(max_pts)
        self.region_cap = int(region_cap) if region_cap else self.n_max * self.max_pts
        self.device = torch.device("cuda", comm.ctx.device) if device is None else torch.device(device)
        self.depth = max(1, int(depth))
        n_out = max(int(n_out), self.depth)
        self.wire_format = wire_format or ("compact" if compact else "exact")
        fmt_code, self.record_bytes = capi.WIRE_FORMATS[self.wire_format]
        self.out = [torch.empty((comm.world, self.region_cap, self.record_bytes), dtype=torch.uint8, device=self.device)
                    for _ in range(n_out)]
        self.expand = bool(expand)
        if self.expand and self.wire_format != "trimmed":
            raise ValueError("expand=True goes with wire_format='trimmed'")
        # expanded regions (SiftPoint records), one per output buffer
        self.out_exact = [torch.empty((comm.world, self.region_cap, SIFT_POINT_BYTES), dtype=torch.uint8, device=self.device)
                          for _ in range(n_out)] if self.expand else None
        comm.set_wire_format(fmt_code)
        comm.reserve(self.n_max, self.depth, self.region_cap)
        if fixed_size:
            comm.set_fixed_size(True)
        self.k = 0
        self._inflight = []  # output buffers of the begins not yet finished, oldest first
        # the producer's tensors are read by the pack that begin() ENQUEUES on the communicator's stream -- a stream
        # torch's allocator knows nothing about -- so they are kept alive here until an event behind that pack has fired
        h = comm.ctx.stream_handle()
        self._stream = torch.cuda.ExternalStream(h, device=self.device) if h else torch.cuda.default_stream(self.device)
        self._held = []  # (event after the pack, (points, counts))

    def begin(self, points, counts, producer=None):
        assert points.is_cuda and points.is_contiguous() and counts.is_cuda and counts.dtype == torch.int32
        n = int(points.shape[0])
        buf = self.out[self.k % len(self.out)]
        self.k += 1
        self.comm.allgatherv_begin(points.data_ptr(), counts.data_ptr(), n, self.max_pts, self.n_max, buf.data_ptr(),
                                   self.region_cap, producer=producer)
        packed = torch.cuda.Event()
        packed.record(self._stream)
        self._held = [(e, t) for e, t in self._held if not e.query()] + [(packed, (points, counts))]
        self._inflight.append((buf, self.out_exact[(self.k - 1) % len(self.out)] if self.expand else None))
        return packed  # after this event the caller may overwrite points / counts (stream.wait_event(packed))

    def finish(self):
        buf, exact = self._inflight.pop(0)
        counts, totals = self.comm.allgatherv_finish()
        if exact is not None:  # expand on arrival: behind the exchange on the communicator's stream
            self.comm.expand_gathered(buf.data_ptr(), self.region_cap, totals, exact.data_ptr())
            return counts, exact, totals
        return counts, buf, totals

    def gather(self, points, counts, producer=None):
        self.begin(points, counts, producer)
        return self.finish()

    def close(self):
        """Waits for the packs still reading producer tensors, then lets go of them."""
        for e, _ in self._held:
            e.synchronize()
        self._held = []

    @staticmethod
    def regions(gathered, totals):
        """List (over ranks) of [totals[r], 588] uint8 views of one gathered buffer."""
        return [gathered[r, : int(totals[r])] for r in range(gathered.shape[0])]


def shard_range(n_total, rank, world):
    """Contiguous block sharding of a batch: images [lo, hi) live on `rank` (64 per GPU for 512 over 8)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_points(points, counts, max_pts):
    """[n, max_pts, 588] uint8 + raw counters -> ([sum(valid), 588] packed records, valid counts [n])."""
    valid = torch.clamp(counts.to(torch.int64), min=0, max=max_pts)
    mask = torch.arange(max_pts, device=points.device)[None, :] < valid[:, None]
    return points[mask], valid.to(torch.int32)


class GatherTicket:
    """First half of an all-gatherv of SiftData: the counts of every rank are on their way to pinned host memory."""

    __slots__ = ("points", "counts", "max_pts", "group", "all_counts", "host_counts", "ready", "world", "rank")


def begin_allgather(points, counts, max_pts, group=None, n_images_max=None):
    """Phase 1 (asynchronous, no host wait): exchange the per-image counts and start copying them to pinned host
    memory.  `n_images_max`: the largest number of images any rank holds; None = find out with one more (blocking)
    exchange.  Call on the stream the exchange should run on, after that stream has been made to wait for `points`."""
    t = GatherTicket()
    t.points, t.counts, t.max_pts, t.group = points, counts, max_pts, group
    t.world, t.rank = dist.get_world_size(group), dist.get_rank(group)
    valid = torch.clamp(counts.to(torch.int64), min=0, max=max_pts).to(torch.int32)
    if n_images_max is None:
        n_local = torch.tensor([valid.numel()], dtype=torch.int32, device=points.device)
        n_all = [torch.zeros_like(n_local) for _ in range(t.world)]
        dist.all_gather(n_all, n_local, group=group)
        n_images_max = int(torch.stack(n_all).max().item())
    n_max = max(int(n_images_max), valid.numel())
    padded_counts = torch.zeros(n_max, dtype=torch.int32, device=points.device)
    padded_counts[: valid.numel()] = valid
    t.all_counts = torch.zeros((t.world, n_max), dtype=torch.int32, device=points.device)
    dist.all_gather_into_tensor(t.all_counts.view(-1), padded_counts, group=group)
    if points.is_cuda:
        t.host_counts = torch.empty((t.world, n_max), dtype=torch.int32, pin_memory=True)
        t.host_counts.copy_(t.all_counts, non_blocking=True)
        t.ready = torch.cuda.Event()
        t.ready.record()
    else:
        t.host_counts, t.ready = t.all_counts, None
    return t


def finish_allgather(t, method="p2p", packer=None):
    """Phase 2: waits for the ticket's counts (an event that has long fired if a step of other work was enqueued in
    between), allocates the result, packs the local shard straight into its place and exchanges the shards.
    Returns (all_counts, gathered, offsets) as allgather_siftdata does."""
    if t.ready is not None:
        t.ready.synchronize()
    points, counts, max_pts, group, world, rank = t.points, t.counts, t.max_pts, t.group, t.world, t.rank
    per_rank = t.host_counts.to(torch.int64).sum(dim=1)
    offsets = torch.zeros(world + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(per_rank, 0)
    total = int(offsets[-1])
    gathered = torch.empty((total, SIFT_POINT_BYTES), dtype=torch.uint8, device=points.device)
    mine = gathered[int(offsets[rank]): int(offsets[rank + 1])]
    # `packer(points, counts, max_pts, out)`: BatchExtractor.make_packer runs one HIP kernel that writes the valid
    # records of all images back to back into `out` -- the local shard is packed straight into its place in
    # `gathered` and sent from there (no staging copy); the torch expression below is the host/gloo form
    if packer is not None:
        packer(points, counts, max_pts, mine)
    else:
        mine.copy_(pack_points(points, counts, max_pts)[0])
    packed = mine

    if method == "padded":
        biggest = int(per_rank.max()) if world else 0
        send = torch.zeros((max(biggest, 1), SIFT_POINT_BYTES), dtype=torch.uint8, device=points.device)
        send[: packed.shape[0]] = packed
        recv = torch.empty((world, max(biggest, 1), SIFT_POINT_BYTES), dtype=torch.uint8, device=points.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
        for r in range(world):
            gathered[int(offsets[r]): int(offsets[r + 1])] = recv[r, : int(per_rank[r])]
    elif method == "p2p":
        ops = []
        for step in range(1, world):
            dst = (rank + step) % world
            src = (rank - step) % world
            if packed.shape[0] > 0:
                ops.append(dist.P2POp(dist.isend, packed, _global_rank(dst, group), group))
            if int(per_rank[src]) > 0:
                ops.append(dist.P2POp(dist.irecv, gathered[int(offsets[src]): int(offsets[src + 1])],
                                      _global_rank(src, group), group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    else:
        raise ValueError("unknown method %r" % method)
    return t.all_counts, gathered, offsets


def allgather_siftdata(points, counts, max_pts, group=None, method="p2p", packer=None, n_images_max=None):
    """All-gatherv of SiftData (both phases back to back; a pipelined caller uses begin_allgather /
    finish_allgather with a step of other work in between, which removes every host wait from its critical path).

    points : uint8 [n_local, max_pts, 588] on this rank; counts : int32 [n_local] raw counters.
    Returns (all_counts int32 [world, n_max], gathered uint8 [total, 588], offsets int64 [world + 1]) where
    rank r's records occupy gathered[offsets[r]:offsets[r+1]] in image order, and all_counts[r, i] is the
    number of points of its i-th image (rows are padded with 0 when ranks hold different image counts).
    """
    return finish_allgather(begin_allgather(points, counts, max_pts, group, n_images_max), method, packer)


def _global_rank(group_rank, group):
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def split_gathered(all_counts, gathered, offsets):
    """Host view: list (over ranks) of lists (over images) of packed uint8 record arrays."""
    out = []
    ac = all_counts.cpu().numpy()
    g = gathered.cpu().numpy()
    for r in range(ac.shape[0]):
        pos = int(offsets[r])
        imgs = []
        for c in ac[r]:
            imgs.append(g[pos: pos + int(c)])
            pos += int(c)
        out.append(imgs)
    return out
