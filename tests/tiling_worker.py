"""Worker of tests/test_dist_gloo.py::test_strip_tiling_*: one rank of a gloo job that runs the strip-tiling HOST
logic (plan, per-octave halo exchange, ownership, merge) with the CPU oracle standing in for the GPU kernels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from cusift_amd import synth  # noqa: E402
from cusift_amd.dist import allgather_siftdata, split_gathered  # noqa: E402
from cusift_amd.tiling import StripPlan, exchange_halos, octave_blurs  # noqa: E402
from oracle_binding import SIFT_POINT_DTYPE, Oracle  # noqa: E402


class CpuBands:
    """The data-movement half of StripExtractor on CPU tensors (same _views contract)."""

    def __init__(self, rank, world, plan):
        self.rank, self.world, self.plan = rank, world, plan
        self.bands = [torch.zeros((plan.band(rank, o)[1] - plan.band(rank, o)[0], plan.pitch[o]), dtype=torch.float32)
                      for o in range(plan.n_oct)]

    def _views(self, o):
        pl = self.plan
        a, b = pl.own(self.rank, o)
        lo, hi = pl.band(self.rank, o)
        t, hal = self.bands[o], pl.halo
        return (t[a - lo: a - lo + hal] if self.rank > 0 else None,
                t[b - lo - hal: b - lo] if self.rank < self.world - 1 else None,
                t[0: a - lo] if self.rank > 0 else None,
                t[b - lo: hi - lo] if self.rank < self.world - 1 else None)


def main():
    out_path = sys.argv[1]
    W, H, n_oct, thresh = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    o_ = Oracle()
    plan = StripPlan(W, H, world, n_oct)
    img = synth.tile(99, W, H)
    cb = CpuBands(rank, world, plan)
    a0, b0 = plan.own(rank, 0)
    lo0, _ = plan.band(rank, 0)
    cb.bands[0][a0 - lo0: b0 - lo0, :W] = torch.from_numpy(img[a0:b0])
    for o in range(n_oct):
        if o > 0:
            # ScaleDown of the previous band; its rows are exact except next to the band's own edges, and the owned
            # rows of this octave are >= HALO/2 - 2 rows away from those
            plo, _ = plan.band(rank, o - 1)
            down = o_.scale_down(cb.bands[o - 1].numpy(), plan.w[o - 1], cb.bands[o - 1].shape[0])
            a, b = plan.own(rank, o)
            lo, _ = plan.band(rank, o)
            cb.bands[o][a - lo: b - lo, : plan.w[o]] = torch.from_numpy(down[a - plo // 2: b - plo // 2, : plan.w[o]].copy())
        if world > 1:
            exchange_halos(cb, o)
    # detection per band with the oracle, centres restricted to owned rows, rows translated to global coordinates
    blur = octave_blurs(0.0, n_oct)
    max_pts = 8192
    pts = np.zeros(max_pts, dtype=SIFT_POINT_DTYPE)
    n = 0
    for o in reversed(range(n_oct)):
        a, b = plan.own(rank, o)
        lo, hi = plan.band(rank, o)
        band = cb.bands[o].numpy()
        dog = o_.laplace_multi(band, plan.w[o], hi - lo, blur[o])
        cand, c = o_.find_points_multi(dog, plan.w[o], hi - lo, thresh, 10.0, float(2 ** o), max_pts)
        cand = cand[:c]
        yi = np.rint(cand["coords2D"][:, 1]).astype(int) + lo  # integer detection row (|pdy| <= 0.5 almost always)
        keep = cand[(yi >= a) & (yi < b)].copy()
        keep["coords2D"][:, 1] += lo
        pts[n: n + len(keep)] = keep
        n += len(keep)
    tp = torch.from_numpy(pts.view(np.uint8).reshape(1, max_pts, 588))
    ac, ga, off = allgather_siftdata(tp, torch.tensor([n], dtype=torch.int32), max_pts)
    merged = np.concatenate([m for r in split_gathered(ac, ga, off) for m in r]).view(SIFT_POINT_DTYPE).reshape(-1)
    np.savez(out_path + ".rank%d.npz" % rank, merged=merged.view(np.uint8),
             **{"band%d" % o: cb.bands[o].numpy() for o in range(n_oct)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
