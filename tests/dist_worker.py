"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job on CPU.

Exercises the N>1 host path (sharding + all-gatherv of SiftData) with the CPU oracle standing in for
the GPU extraction (tests may use the oracle; the product never does).  Writes its result to argv[1].
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from cusift_amd import synth  # noqa: E402
from cusift_amd.dist import (allgather_siftdata, begin_allgather, finish_allgather, shard_range,  # noqa: E402
                             split_gathered)
from oracle_binding import SIFT_POINT_DTYPE, Oracle  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    result = {}
    if mode == "random":
        # ragged: different image counts per rank, empty images, an empty rank
        g = torch.Generator().manual_seed(100 + rank)
        n_local, max_pts = (0 if rank == 1 and world > 2 else 3 + rank), 10
        pts = torch.randint(0, 255, (n_local, max_pts, 588), dtype=torch.uint8, generator=g)
        cnt = torch.tensor([(i * 5 + rank * 3) % 14 for i in range(n_local)], dtype=torch.int32)  # some > max_pts, some 0
        for method in ("p2p", "padded"):
            ac, ga, off = allgather_siftdata(pts, cnt, max_pts, method=method)
            result[method] = dict(counts=ac.numpy(), gathered=ga.numpy(), offsets=off.numpy())
        result["local_pts"] = pts.numpy()
        result["local_cnt"] = cnt.numpy()
    elif mode == "pipelined":
        # the order bench.py uses: phase 1 of step i+1 is issued before phase 2 of step i (two tickets in flight),
        # with the image-count hint that skips the first exchange; every step must equal the one-shot form
        max_pts, n_local = 12, 4
        steps = []
        for k in range(3):
            g = torch.Generator().manual_seed(1000 * k + rank)
            pts = torch.randint(0, 255, (n_local, max_pts, 588), dtype=torch.uint8, generator=g)
            cnt = torch.tensor([(7 * i + 5 * rank + 3 * k) % 16 for i in range(n_local)], dtype=torch.int32)
            steps.append((pts, cnt))
        tickets = [begin_allgather(p, c, max_pts, n_images_max=n_local) for p, c in steps[:2]]
        outs = [finish_allgather(tickets[0])]
        tickets.append(begin_allgather(steps[2][0], steps[2][1], max_pts, n_images_max=n_local))
        outs += [finish_allgather(tickets[1]), finish_allgather(tickets[2])]
        for k, ((pts, cnt), (ac, ga, off)) in enumerate(zip(steps, outs)):
            ac1, ga1, off1 = allgather_siftdata(pts, cnt, max_pts)
            result["step%d" % k] = dict(counts=ac.numpy(), gathered=ga.numpy(), offsets=off.numpy(),
                                        counts1=ac1.numpy(), gathered1=ga1.numpy(), offsets1=off1.numpy())
    else:
        # the real thing at small scale: a batch of 5 images sharded over the ranks, oracle extraction,
        # all-gatherv, every rank ends up with the same merged SiftData in global image order
        n_total, w, h, max_pts = 5, 160, 120, 512
        lo, hi = shard_range(n_total, rank, world)
        o = Oracle()
        kw = dict(num_octaves=3, init_blur=0.0, peak_thresh=1.0, max_pts=max_pts)
        pts = np.zeros((hi - lo, max_pts), dtype=SIFT_POINT_DTYPE)
        cnt = np.zeros(hi - lo, dtype=np.int32)
        for i in range(lo, hi):
            p = o.extract(synth.tile(1000 + i, w, h), **kw)
            pts[i - lo, : len(p)] = p
            cnt[i - lo] = len(p)
        tp = torch.from_numpy(pts.view(np.uint8).reshape(hi - lo, max_pts, 588))
        ac, ga, off = allgather_siftdata(tp, torch.from_numpy(cnt), max_pts)
        per = split_gathered(ac, ga, off)
        merged = [img for r in range(world) for img in per[r][: shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]]]
        result["merged_counts"] = np.array([len(m) for m in merged])
        result["merged_bytes"] = np.concatenate([m.reshape(-1) for m in merged]) if merged else np.zeros(0, np.uint8)
    np.savez(out_path + ".rank%d.npz" % rank, **{k: v for k, v in _flatten(result).items()})
    dist.barrier()
    dist.destroy_process_group()


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


if __name__ == "__main__":
    main()
