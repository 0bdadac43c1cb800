#!/usr/bin/env python3
"""Where does a context's side stream land?  `n` extraction streams (a PipelinedExtractor, as bench.py builds it) are
used first, then each of its extractors is driven as a lone caller (concurrent_batches = 1, 64 x 1080p): the context
creates its side stream after the concurrency probe (policy CUSIFT_POLICY_SIDE_STREAM = 2; the probe's timings are
printed by a -DCUSIFT_LAB build with CUSIFT_SIDE_DEBUG=1) and forks.

    GPU_MAX_HW_QUEUES=8 python tools/probe_side_stream.py 4

MI355X, ROCm 7.2: with 4 streams in use and 8 hardware queues the first candidate shares a command-processor pipe with
the context's stream (probe chain 100 -> 150 us; a fork over it runs 1.88 ms per batch instead of 1.35) and is
rejected, the second is kept; with 3 streams and 4 queues the first candidate IS the context's queue (100 -> 222 us).
"""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES","8")
os.environ["CUSIFT_SIDE_DEBUG"]="1"
sys.path.insert(0,'.')
import torch
from cusift_amd import capi, synth
from cusift_amd.batch import BatchExtractor, PipelinedExtractor
B,w,h=64,1920,1080
base=[synth.tile(1000+i,w,h,1.0) for i in range(4)]
imgs=np.stack([base[i%4] for i in range(B)])
kw=dict(num_octaves=5,init_blur=1.0,peak_thresh=3.0,max_pts=32768)
ns=int(sys.argv[1])
pipe=PipelinedExtractor(B,w,h,n_streams=ns,n_slots=1,**kw)
exs=pipe.extractors
d=exs[0].images_from_numpy(imgs)
def lone(ex,n=30):
    ex.params.concurrent_batches=1
    ex.ctx.set_policy(capi.POLICY_SIDE_STREAM, 2)
    for _ in range(3): ex.extract(d)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): ex.extract(d)
    torch.cuda.synchronize(); r=(time.perf_counter()-t0)/n*1e3
    ex.params.concurrent_batches=ns
    return r
for i,ex in enumerate(exs):
    print("queues %s, %d streams: extractor %d as lone caller %.4f ms, forks %d" % (os.environ["GPU_MAX_HW_QUEUES"], ns, i, lone(ex), ex.ctx.forks()), flush=True)
