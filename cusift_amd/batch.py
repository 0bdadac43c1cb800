"""Batched SIFT extraction on torch device tensors.

torch is plumbing here (HBM allocations, the stream, torch.distributed); every kernel is launched by
libcusift_amd.so through the C ABI on the stream this object was created on.
"""
import numpy as np
import torch

from . import capi


class BatchExtractor:
    """Extracts SiftData for a batch of equally sized, HBM-resident float32 images.

    The batch form of SiftData::Extract (cuSIFT.cu:61-120): `extract()` only enqueues work -- no
    allocation, no host read-back -- and returns device tensors:
        points  uint8  [n, max_pts, 588]   (SiftPoint records, cuSIFT.h:10-30)
        counts  int32  [n]                 raw counters; valid points = min(count, max_pts)
    """

    def __init__(self, n_images, w, h, params=None, device=None, pitch=None, n_slots=1, **param_overrides):
        if not torch.cuda.is_available():
            raise capi.CusiftError("BatchExtractor needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n, self.w, self.h = int(n_images), int(w), int(h)
        self.pitch = capi.ialign_up(self.w, 128) if pitch is None else int(pitch)
        self.params = params if params is not None else capi.default_params(**param_overrides)
        self.max_pts = self.params.max_pts
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream()
            self.ctx = capi.Context(self.device.index, stream=self.stream.cuda_stream)
            self.ctx.reserve(self.n, self.w, self.h, self.params)
            # n_slots > 1: output ring, so that a consumer (D2H copy, all-gatherv on another stream) can still
            # read step i's SiftData while step i+1 is being extracted
            self.slots = [(torch.zeros((self.n, self.max_pts, capi.SIFT_POINT_BYTES), dtype=torch.uint8,
                                       device=self.device),
                           torch.zeros((self.n,), dtype=torch.int32, device=self.device)) for _ in range(n_slots)]
            self.points, self.counts = self.slots[0]

    def images_from_numpy(self, imgs):
        """(n, h, w) float32 host array -> pitched device tensor (n, h, pitch)."""
        imgs = np.asarray(imgs, dtype=np.float32)
        assert imgs.shape == (self.n, self.h, self.w), imgs.shape
        dev = torch.zeros((self.n, self.h, self.pitch), dtype=torch.float32, device=self.device)
        dev[:, :, : self.w] = torch.from_numpy(imgs).to(self.device)
        return dev

    def extract(self, d_imgs, slot=0):
        assert d_imgs.is_cuda and d_imgs.dtype == torch.float32 and d_imgs.is_contiguous()
        assert tuple(d_imgs.shape) == (self.n, self.h, self.pitch), tuple(d_imgs.shape)
        self.points, self.counts = self.slots[slot]
        self.ctx.extract_batch(d_imgs.data_ptr(), self.n, self.w, self.h, self.pitch, self.h * self.pitch,
                               self.params, self.points.data_ptr(), self.counts.data_ptr())
        return self.points, self.counts

    def make_packer(self, stream=None):
        """A `packer(points, counts, max_pts)` for cusift_amd.dist.allgather_siftdata that runs cusift_pack_points on
        `stream` (a torch.cuda.Stream; default: this extractor's stream) through a context of its own that borrows
        that stream -- so the exchange of step i can be packed and sent on a side stream while step i+1 is extracted."""
        stream = self.stream if stream is None else stream
        ctx = self.ctx if stream == self.stream else capi.Context(self.device.index, stream=stream.cuda_stream)
        self._packer_ctxs = getattr(self, "_packer_ctxs", []) + [ctx]

        def packer(points, counts, max_pts, out=None):
            """Valid records of all images back to back.  out=None: returns (packed, valid counts) like
            cusift_amd.dist.pack_points (one host read-back for the size); out=<uint8 [total, 588] tensor of exactly
            the valid total>: packs into it, no read-back, returns None."""
            n = points.shape[0]
            if n > 256 or not points.is_cuda:  # kMaxFlatImages
                from .dist import pack_points
                packed, valid = pack_points(points, counts, max_pts)
                if out is None:
                    return packed, valid
                out.copy_(packed)
                return None
            with torch.cuda.stream(stream):
                if out is None:
                    valid = torch.clamp(counts, max=max_pts)
                    total = int(valid.sum().item())  # the one host read-back of the packing
                    out_t = torch.empty((total, capi.SIFT_POINT_BYTES), dtype=torch.uint8, device=points.device)
                else:
                    assert out.is_contiguous() and out.dtype == torch.uint8 and out.shape[1] == capi.SIFT_POINT_BYTES
                    out_t, total = out, out.shape[0]
                if total > 0:
                    ctx.pack_points(points.data_ptr(), counts.data_ptr(), n, max_pts, out_t.data_ptr(), total, None)
            return (out_t, valid) if out is None else None

        return packer

    def valid_counts(self):
        return torch.clamp(self.counts, max=self.max_pts)

    def to_host(self):
        """Blocking: list of numpy SiftPoint arrays, one per image."""
        cnt = self.valid_counts().cpu().numpy()
        out = []
        for i in range(self.n):
            raw = self.points[i, : int(cnt[i])].cpu().numpy()
            out.append(raw.view(capi.SIFT_POINT_DTYPE).reshape(-1))
        return out

    def close(self):
        self.ctx.close()


class PipelinedExtractor:
    """Consecutive batches on `n_streams` HIP streams, one BatchExtractor (context, arena, output slots) each.

    A single launch sequence leaves the chip part-idle at times: the ScaleDown chain is HBM-bound while everything
    else is VALU-bound, and the last waves of every detection / description launch run on a nearly empty device.
    With several batches in flight those gaps are filled by the other batches' kernels, and the detection can use tall
    row chunks (cusift_params.concurrent_batches, set here to the stream count).  64 x 1080p on MI355X: 1.64 ms per
    batch on one stream, 1.44 on two, 1.41 on four (1.38 with the chunk hint).  Results are those of BatchExtractor.

        pipe = PipelinedExtractor(64, 1920, 1080, n_streams=4, num_octaves=5, init_blur=1.0, peak_thresh=3.0)
        for frames in source:                       # frames: float32 device tensor [n, h, pitch]
            points, counts, done = pipe.submit(frames)
            ...                                     # consume after `done` (an event) -- e.g. other_stream.wait_event(done)
    The output tensors of a submit are reused by the submit `n_streams * n_slots` calls later.
    """

    def __init__(self, n_images, w, h, n_streams=4, n_slots=1, device=None, **kw):
        if not torch.cuda.is_available():
            raise capi.CusiftError("PipelinedExtractor needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(max(1, n_streams) - 1)]
            kw.setdefault("concurrent_batches", len(self.streams))  # scheduling hint: other batches fill launch tails
            self.extractors = []
            for st in self.streams:
                with torch.cuda.stream(st):
                    self.extractors.append(BatchExtractor(n_images, w, h, n_slots=n_slots, device=self.device, **kw))
        self.n_slots = n_slots
        self.submitted = 0
        first = self.extractors[0]
        self.n, self.w, self.h, self.pitch, self.max_pts, self.params = (first.n, first.w, first.h, first.pitch,
                                                                         first.max_pts, first.params)

    def images_from_numpy(self, imgs):
        return self.extractors[0].images_from_numpy(imgs)

    def submit(self, d_imgs, ready=None):
        """Enqueue one batch on the next stream.  `ready`: an event after which d_imgs may be read (None: the caller
        has made sure already, e.g. by a device synchronisation after the upload).
        Returns (points, counts, done_event)."""
        k = self.submitted
        self.submitted += 1
        e = k % len(self.streams)
        st = self.streams[e]
        with torch.cuda.stream(st):
            if ready is not None:
                st.wait_event(ready)
            pts, cnt = self.extractors[e].extract(d_imgs, slot=(k // len(self.streams)) % self.n_slots)
            done = torch.cuda.Event()
            done.record(st)
        return pts, cnt, done

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for x in self.extractors:
            x.close()
