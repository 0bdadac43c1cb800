#!/usr/bin/env python3
"""Experiment: does an initialised RCCL communicator (torch.distributed, backend nccl) slow unrelated kernels down?
Times the fused detection and the keypoint kernel of one 64x1080p batch before and after init_process_group."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cusift_amd import synth  # noqa: E402
from cusift_amd.batch import BatchExtractor  # noqa: E402

B, w, h = 64, 1920, 1080
ex = BatchExtractor(B, w, h, num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
d = ex.images_from_numpy(np.stack([synth.tile(1000, w, h, 1.0)] * B))
torch.cuda.synchronize()


def measure(tag):
    for _ in range(3):
        ex.extract(d)
    torch.cuda.synchronize()
    ex.ctx.timing_enable(True)
    ex.ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(10):
        ex.extract(d)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    t = ex.ctx.timing_read()
    ex.ctx.timing_enable(False)
    print("%-28s wall %.4f ms  scale_down %.4f  detect %.4f  describe %.4f" %
          (tag, wall, t["scale_down"][0] / 10, t["detect_multi"][0] / 10, t["describe_all"][0] / 10), flush=True)


measure("before init_process_group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29542")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
if os.environ.get("EXP_LAZY"):
    dist.init_process_group("nccl", rank=0, world_size=1)
else:
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
measure("after init (no collective yet)")
x = torch.ones(16, device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
measure("after the first collective")
dist.destroy_process_group()
measure("after destroy_process_group")
