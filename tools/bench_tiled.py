#!/usr/bin/env python3
"""BASELINE configs[4]: one 8192x8192 image, strip-tiled over P ranks with per-octave halo exchange.

  python tools/bench_tiled.py                      # one GPU: whole image vs P=8 virtual ranks (halo exchange = copies)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_tiled.py   # real 8 GPUs

Prints one JSON line (informational; the driver's headline bench is bench.py).
"""
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the first HIP call: ROCr reads it at hsa_init

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from cusift_amd import capi, synth
    from cusift_amd.dist import SiftGatherer, make_comm
    from cusift_amd.tiling import StripExtractor, run_distributed, run_virtual

    W = H = int(os.environ.get("TILED_SIZE", "8192"))
    steps = int(os.environ.get("TILED_STEPS", "5"))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    prm = capi.default_params(num_octaves=int(os.environ.get("TILED_OCTAVES", "5")), init_blur=1.0, peak_thresh=3.0,
                              max_pts=1 << 19)
    img = synth.tile(4242, W, H, preblur=1.0)
    out = {"workload": "single %dx%d image, 5 octaves, initBlur=1.0, thresh=3.0" % (W, H)}

    if world > 1:
        dist.init_process_group("nccl", device_id=dev)  # carries the communicator id + the barrier only
        # the halo / row exchanges and the all-gatherv go through the C ABI's communicator (RCCL from C++)
        ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
        comm = make_comm(ctx)
        ext = StripExtractor(rank, world, W, H, prm, device=dev, comm=comm)
        b = ext.plan.bounds
        strip = torch.from_numpy(img[b[rank]:b[rank + 1]]).to(dev)
        gat = SiftGatherer(comm, 1, ext.max_pts, region_cap=ext.max_pts, device=dev)
        for it in range(steps + 2):
            if it == 2:
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            pts, cnt = run_distributed(ext, strip)
            ac, ga, totals = gat.gather(pts, cnt)
        ext.check()
        dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if rank == 0:
            out.update({"n_gpus": world, "ms_per_image": round(dt * 1e3, 3), "Mpix_per_s": round(W * H / dt / 1e6, 1),
                        "keypoints": int(sum(int(t) for t in totals)), "collapse_octave": ext.plan.collapse,
                        "mode": "distributed strips + halo exchange + all-gatherv (C ABI over RCCL)"})
            print(json.dumps(out), flush=True)
        comm.close()
        ctx.close()
        dist.destroy_process_group()
        return

    # one GPU: the whole image through the batch driver (n = 1) ...
    from cusift_amd.batch import BatchExtractor

    ex = BatchExtractor(1, W, H, params=prm)
    d_img = ex.images_from_numpy(img[None])
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        ex.extract(d_img)
    torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / steps
    n_whole = int(ex.valid_counts().sum().item())
    # ... and P = 8 virtual ranks run one after the other on the same GPU (per-rank time ~ total / 8)
    P = 8
    full = torch.from_numpy(img).to(dev)
    rows = H // P
    exts = [StripExtractor(k, P, W, H, prm, device=dev) for k in range(P)]
    strips = [full[k * rows:(k + 1) * rows] for k in range(P)]
    for it in range(steps + 1):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        parts = run_virtual(exts, strips)
    torch.cuda.synchronize()
    virt = (time.perf_counter() - t0) / steps
    out.update({"n_gpus": 1, "whole_image_ms": round(whole * 1e3, 3), "whole_image_Mpix_per_s": round(W * H / whole / 1e6, 1),
                "keypoints_whole": n_whole, "virtual_ranks": P, "virtual_8rank_total_ms": round(virt * 1e3, 3),
                "virtual_per_rank_ms_estimate": round(virt * 1e3 / P, 3), "keypoints_tiled": int(sum(len(p) for p in parts)),
                "note": "virtual ranks share one GPU and include the D2H of their results; per-rank estimate = total/8"})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
