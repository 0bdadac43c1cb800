#!/usr/bin/env python3
"""Round 6, VERDICT item 6: where do the last 13 % of the upload link go?  The host-to-host leg uploads 64 x 1080p 8-bit
frames (132.7 MB) per step at 0.87 of what the same copy reaches alone.  Three things are tried, each alone, against the
plain arrangement (one upload stream, coherent pinned memory, one 132.7 MB copy per step, 99 MB of records going the other
way at the same time):
  a. two upload streams with half a batch each (two SDMA engines);
  b. the source in NON-COHERENT pinned memory (hipHostMallocNonCoherent) / NUMA-user / write-combined;
  c. copy sizes: the batch as 1, 2, 4, 8 copies.
Prints one JSON line: GB/s of the upload alone and beside a D2H stream carrying 99 MB per step, per arrangement."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FLAGS = {"default (coherent)": 0x0, "non_coherent": 0x80000000, "numa_user": 0x20000000, "write_combined": 0x4,
         "portable": 0x1}


def main():
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    up_bytes = 64 * 1920 * 1080
    down_bytes = 99 * 1000 * 1000
    d_up = [torch.empty(up_bytes, dtype=torch.uint8, device=dev) for _ in range(3)]
    d_down = torch.zeros(down_bytes, dtype=torch.uint8, device=dev)
    h_down = torch.empty(down_bytes, dtype=torch.uint8).pin_memory()
    s_up = [torch.cuda.Stream() for _ in range(2)]
    s_down = torch.cuda.Stream()
    out = {}

    def measure(h_ptr, pieces, streams, with_down, reps=24):
        def go(n):
            for r in range(n):
                dst = d_up[r % 3].data_ptr()
                chunk = up_bytes // pieces
                for p in range(pieces):
                    st = s_up[p % streams]
                    hip.hipMemcpyAsync(dst + p * chunk, h_ptr + p * chunk, chunk, 1, C.c_void_p(st.cuda_stream))
                if with_down:
                    with torch.cuda.stream(s_down):
                        h_down.copy_(d_down, non_blocking=True)
        go(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(reps)
        torch.cuda.synchronize()
        return up_bytes * reps / (time.perf_counter() - t0) / 1e9

    for name, flag in FLAGS.items():
        p = C.c_void_p()
        rc = hip.hipHostMalloc(C.byref(p), up_bytes, flag)
        if rc != 0:
            out[name] = {"error": "hipHostMalloc rc %d" % rc}
            continue
        C.memset(p.value, 7, up_bytes)  # touch
        row = {}
        for pieces, streams in ((1, 1), (2, 1), (2, 2), (4, 2), (8, 2), (8, 1)):
            key = "%d copies on %d stream%s" % (pieces, streams, "s" if streams > 1 else "")
            row[key] = {"alone_GBps": round(measure(p.value, pieces, streams, False), 2),
                        "beside_99MB_d2h_GBps": round(measure(p.value, pieces, streams, True), 2)}
        out[name] = row
        hip.hipHostFree(p)
    # d. a TINY device-to-host copy on a third stream (the records' offsets: 260 bytes per step, which the host waits for
    #    before it sizes the records' copy) beside the two large ones: how long does the host wait for it?
    p = C.c_void_p()
    hip.hipHostMalloc(C.byref(p), up_bytes, 0)
    s_tiny = torch.cuda.Stream()
    d_tiny = torch.zeros(65, dtype=torch.int32, device=dev)
    h_tiny = torch.zeros(65, dtype=torch.int32).pin_memory()
    for label, with_up, with_down in (("alone", False, False), ("beside the upload", True, False),
                                      ("beside upload and 99 MB d2h", True, True)):
        waits = []
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for r in range(16):
            if with_up:
                hip.hipMemcpyAsync(d_up[r % 3].data_ptr(), p.value, up_bytes, 1, C.c_void_p(s_up[0].cuda_stream))
            if with_down:
                with torch.cuda.stream(s_down):
                    h_down.copy_(d_down, non_blocking=True)
            with torch.cuda.stream(s_tiny):
                h_tiny.copy_(d_tiny, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(s_tiny)
            t0 = time.perf_counter()
            ev.synchronize()
            waits.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        out.setdefault("tiny_d2h_wait_ms", {})[label] = {"median": round(float(np.median(waits)), 4),
                                                        "max": round(float(np.max(waits)), 4)}
    hip.hipHostFree(p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
