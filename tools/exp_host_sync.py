#!/usr/bin/env python3
"""Experiment: which part of the per-step all-gatherv of SiftData costs extraction throughput on ONE rank?
All variants run on a side stream that first waits for the step's extraction (two extraction streams alternate)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cusift_amd import capi, synth  # noqa: E402
from cusift_amd.batch import PipelinedExtractor  # noqa: E402
from cusift_amd.dist import begin_allgather, finish_allgather  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
B, w, h = 64, 1920, 1080
pipe = PipelinedExtractor(B, w, h, n_streams=2, n_slots=3, num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
d = pipe.images_from_numpy(np.stack([synth.tile(1000, w, h, 1.0)] * B))
side = torch.cuda.Stream()
packer = pipe.extractors[0].make_packer(side)
torch.cuda.synchronize()
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
scratch = torch.empty((200000, 588), dtype=torch.uint8, device=dev)
scratch_i = torch.zeros((1, 64), dtype=torch.int32, device=dev)


def run(mode, steps=24):
    pend = []
    for _ in range(6):
        pipe.submit(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pts, cnt, ev = pipe.submit(d)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            if mode == "smallops":
                v = torch.clamp(cnt.to(torch.int64), min=0, max=32768).to(torch.int32)
                z = torch.zeros((1, 64), dtype=torch.int32, device=dev)
                z[0, :64] = v
            elif mode == "op_clamp":
                torch.clamp(cnt, min=0, max=32768)
            elif mode == "op_to64":
                cnt.to(torch.int64)
            elif mode == "op_zeros":
                torch.zeros((1, 64), dtype=torch.int32, device=dev)
            elif mode == "op_slice_assign":
                scratch_i[0, :64] = cnt
            elif mode == "op_zeros_1d_assign":
                z = torch.zeros(64, dtype=torch.int32, device=dev)
                z[:64] = cnt
            elif mode == "allgather":
                out = torch.zeros(64, dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(out, cnt)
            elif mode == "begin":
                begin_allgather(pts, cnt, 32768, n_images_max=B)
            elif mode == "pack":
                packer(pts, cnt, 32768, scratch[:168816])
            elif mode == "empty99":
                torch.empty((168816, 588), dtype=torch.uint8, device=dev)
            elif mode == "full":
                pend.append(begin_allgather(pts, cnt, 32768, n_images_max=B))
                if len(pend) > 1:
                    finish_allgather(pend.pop(0), packer=packer)
    torch.cuda.synchronize()
    print("%-12s %.4f ms/step" % (mode, (time.perf_counter() - t0) / steps * 1e3), flush=True)


for m in ("none", "op_clamp", "op_to64", "op_zeros", "op_slice_assign", "op_zeros_1d_assign", "smallops", "begin", "none"):
    run(m)
dist.destroy_process_group()
