#!/usr/bin/env python3
"""Round 6, VERDICT item 6 (second half): the host-to-host leg reaches 0.87 of the upload link.  Which part of the step
costs the upload its rate?  64 x 1080p 8-bit frames uploaded every step (one stream, pinned source, three device buffers
in flight), plus -- switched on one at a time -- the conversion + extraction on the four-stream pipeline, the pack, the
records' read-back.  Prints one JSON line: H2D GB/s of every combination."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("PROBE_HW_QUEUES", "8"))  # as bench.py: read at HIP initialisation

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from cusift_amd import capi, synth
    from cusift_amd.batch import PipelinedExtractor

    B, w, h = 64, 1920, 1080
    dev = torch.device("cuda", 0)
    prm_kw = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, lowest_scale=0.0, subsampling=1.0,
                  max_pts=32768, tex_frac_bits=8)
    out = {}
    for E in (3, 4):
        pipe = PipelinedExtractor(B, w, h, n_streams=E, n_slots=1, fused_detect=1, **prm_kw)
        ex = pipe.extractors[0]
        np_imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(B)])
        h_src = torch.from_numpy(np.clip(np.rint(np_imgs), 0, 255).astype(np.uint8)).pin_memory()
        n_in = 3
        d_u8 = [torch.empty((B, h, w), dtype=torch.uint8, device=dev) for _ in range(n_in)]
        d_in = [torch.zeros((B, h, pipe.pitch), dtype=torch.float32, device=dev) for _ in range(n_in)]
        h2d = torch.cuda.Stream()
        pack_stream, copy_stream = torch.cuda.Stream(), torch.cuda.Stream()
        cctx = capi.Context(0, stream=pack_stream.cuda_stream)
        cap = 300000
        packed = [torch.empty((cap, 588), dtype=torch.uint8, device=dev) for _ in range(4)]
        offs = [torch.zeros(B + 1, dtype=torch.int32, device=dev) for _ in range(4)]
        h_rec = [torch.empty((cap, 588), dtype=torch.uint8).pin_memory() for _ in range(4)]

        def run(extract, pack, readback, steps=16):
            free = [None] * n_in
            copied = [None] * 4
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                b = i % n_in
                with torch.cuda.stream(h2d):
                    if free[b] is not None:
                        h2d.wait_event(free[b])
                    d_u8[b].copy_(h_src, non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(h2d)
                if not extract:
                    continue
                e = pipe.submitted % E
                with torch.cuda.stream(pipe.streams[e]):
                    pipe.streams[e].wait_event(up)
                    pipe.extractors[e].ctx.u8_to_f32(d_in[b].data_ptr(), pipe.pitch, d_u8[b].data_ptr(), w, h, w, n_images=B)
                pts, cnt, ev = pipe.submit(d_in[b])
                free[b] = ev
                if not pack:
                    continue
                j = i % 4
                with torch.cuda.stream(pack_stream):
                    pack_stream.wait_event(ev)
                    if copied[j] is not None:
                        pack_stream.wait_event(copied[j])
                    cctx.pack_points(pts.data_ptr(), cnt.data_ptr(), B, 32768, packed[j].data_ptr(), cap, offs[j].data_ptr())
                    done = torch.cuda.Event()
                    done.record(pack_stream)
                if readback:
                    with torch.cuda.stream(copy_stream):
                        copy_stream.wait_event(done)
                        h_rec[j][:170000].copy_(packed[j][:170000], non_blocking=True)  # ~100 MB, as the leg's
                        copied[j] = torch.cuda.Event()
                        copied[j].record(copy_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            return round(h_src.numel() * steps / dt / 1e9, 2)

        row = {}
        for label, a in (("upload only", (False, False, False)), ("+ conversion + extraction", (True, False, False)),
                         ("+ pack", (True, True, False)), ("+ records' read-back (the leg)", (True, True, True))):
            run(*a, steps=4)
            row[label] = run(*a)
        out["%d extraction streams" % E] = row
        cctx.close()
        for x in pipe.extractors:
            x.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
