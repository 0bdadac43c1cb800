"""`cpu_baseline`: the CPU oracle (a restatement of the cuSIFT algorithm -- NOT OpenCV) timed on the host cores, and
OpenCV's CPU SIFT if a box has it.  The ONLY place bench.py touches anything under oracle/ -- as the thing timed beside
the number, never as the product."""
import os
import sys
import time

import numpy as np

from .common import ROOT, usable_cpus


def cpu_baseline(w, h, params_kw, preblur, budget_s):
    """The CPU oracle (a restatement of the cuSIFT algorithm -- NOT OpenCV) timed on the host cores: one image per
    thread (the C code releases the GIL), bounded to ~budget_s of wall time."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cusift_amd import synth
    from oracle_binding import Oracle  # checker / baseline only

    cores, usable = usable_cpus()
    # measured on the box (16-CPU quota): 8 / 16 / 24 / 32 / 64 threads -> 102 / 184 / 197 / 189 / 163 Mpix/s
    threads = int(os.environ.get("CUSIFT_CPU_THREADS", "0")) or max(1, min(cores, usable + usable // 2, 48))
    oracle = Oracle()
    imgs = [synth.tile(5000 + i, w, h, preblur) for i in range(threads)]
    # wall-clock median of >= 5 runs after one warm-up.  A run = every thread extracts one image, all at the
    # same time (the C code releases the GIL); runs repeat until the budget is spent.
    counts = [0] * threads

    def one_pass():
        def work(i):
            counts[i] = len(oracle.extract(imgs[i], **params_kw))
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    one_pass()  # warm-up (page-in, first-touch)
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - t_start < budget_s and len(times) < 200):
        times.append(one_pass())
    dt = sum(times)
    n_img = threads * len(times)
    rates = sorted(threads * w * h / t / 1e6 for t in times)
    med = rates[len(rates) // 2]
    out = {
        "value": round(med, 3),
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": "median of %d runs after 1 warm-up, each %d x %dx%d images at once on %d threads (same generator/"
                  "params), %.1f s wall in all; CPU restatement of the cuSIFT algorithm (oracle/sift_oracle.c), not OpenCV"
                  % (len(times), threads, w, h, threads, dt),
        "runs": len(times),
        "spread_mpix_per_s": {"min": round(rates[0], 3), "median": round(med, 3), "max": round(rates[-1], 3)},
        "keypoints_per_s": round(sum(counts) * med * 1e6 / (threads * w * h), 1),
        "host_cores": cores,
        "usable_cpus": usable,
    }
    out["opencv"] = opencv_baseline(imgs[: min(len(imgs), 8)], usable, min(budget_s, 8.0))
    return out


def opencv_baseline(imgs, threads, budget_s):
    """north_star asks for OpenCV's CPU SIFT beside the number (the reference's callers decode with OpenCV,
    test/detector.cpp:19-20).  It is not in this image; if a box has it, it is timed on the same images."""
    try:
        import cv2  # noqa: F401
    except Exception as e:  # ImportError, or a broken binary wheel
        return {"available": False, "note": "opencv: absent (import cv2: %s)" % type(e).__name__}
    try:
        cv2.setNumThreads(int(threads))
        sift = cv2.SIFT_create(0, 3, 0.04, 10, 1.6)
        u8 = [np.clip(i, 0, 255).astype(np.uint8) for i in imgs]
        sift.detectAndCompute(u8[0], None)
        n, kp, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            k, _ = sift.detectAndCompute(u8[n % len(u8)], None)
            kp += len(k)
            n += 1
        dt = time.perf_counter() - t0
        h, w = u8[0].shape
        return {"available": True, "version": cv2.__version__, "threads": int(threads),
                "Mpix_per_s": round(n * w * h / dt / 1e6, 3), "keypoints_per_s": round(kp / dt, 1),
                "note": "cv2.SIFT_create(0,3,0.04,10,1.6).detectAndCompute on the same images (8-bit)"}
    except Exception as e:
        return {"available": False, "note": "opencv: present but SIFT failed (%s)" % e}


def configs0_cpu(hip_keypoints=None, runs=7):
    """BASELINE configs[0] -- the reference's own CPU-runnable case -- on the CPU: the oracle on ONE core over the 640x480
    fixture with the GPU leg's parameters (config_legs["configs[0]"]); as the checker it also compares keypoint counts."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cusift_amd import synth
    from oracle_binding import Oracle  # checker / baseline only

    oracle = Oracle()
    img = synth.fixture_image()
    kw = dict(num_octaves=3, init_blur=0.0, peak_thresh=0.1, edge_thresh=10.0, max_pts=16384)
    n = len(oracle.extract(img, **kw))  # warm-up
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        n = len(oracle.extract(img, **kw))
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    h, w = img.shape
    out = {"workload": "the 640x480 fixture, 3 octaves, initBlur=0, thresh=0.1 (BASELINE configs[0]); CPU restatement of the "
                       "cuSIFT algorithm on one core, median of %d runs" % runs,
           "ms_per_image": round(med * 1e3, 3), "Mpix_per_s": round(w * h / med / 1e6, 3), "cores": 1, "keypoints": int(n)}
    if hip_keypoints is not None:
        out["keypoints_equal_hip"] = bool(int(hip_keypoints) == int(n))
    return out
