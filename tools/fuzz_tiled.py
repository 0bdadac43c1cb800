#!/usr/bin/env python3
"""Randomised sweep of the strip-tiled extraction (csrc/sift_tiled.hip) on ONE GPU: random image sizes, rank counts,
octave counts, halo depths, contents and parameters; the P ranks run as P threads over the test transport
(tests/fake_rccl: cusift_tiled_extract with real halo / row exchanges through the C ABI) or, every third case, as
virtual ranks (device copies).  Run on the GPU box:

    python tools/fuzz_tiled.py [n_cases] [seed]

Per case: the union of the ranks' SiftData must equal the whole-image extraction bit for bit (every field extraction
writes), every rank's list must be coarsest octave first, and the all-gatherv of the rank lists must arrive the same on
every rank.  A halo too shallow for the image's keypoints is a legitimate outcome -- cusift_tiled_check must then flag
it (the case is counted as "flagged", not compared)."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from cusift_amd import capi, synth  # noqa: E402
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer  # noqa: E402
from cusift_amd.dist import SiftGatherer  # noqa: E402
from cusift_amd.tiling import StripExtractor, run_distributed, run_virtual  # noqa: E402
from fake_transport import fake_rccl_path, fake_stats  # noqa: E402
from parity_utils import canonical_order  # noqa: E402

FIELDS = ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f], equal_nan=True) for f in FIELDS)


def run_threads(world, fn):
    capi.comm_use_library(fake_rccl_path())
    errors, results = [], [None] * world
    try:
        uid = capi.comm_unique_id()

        def body(rank):
            try:
                results[rank] = fn(rank, lambda ctx: capi.Comm(ctx, uid, rank, world))
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, e))

        ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(600)
        if any(t.is_alive() for t in ts):
            raise RuntimeError("a rank hung")
    finally:
        capi.comm_use_library(None)
    return results, errors


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    failed = flagged = 0
    for case in range(n_cases):
        P = int(rng.integers(1, 9))
        tiny = rng.integers(0, 6) == 0  # a sixth of the cases: images a few rows / columns large
        W = int(rng.integers(4, 70)) if tiny else int(rng.integers(16, 1400))
        H = int(rng.integers(P, P + 90)) if tiny else int(rng.integers(max(P, 24), 2200))
        n_oct = int(rng.integers(1, 8))
        halo = int(rng.choice([16, 24, 32, 48, 64]))
        blur = float(rng.choice([0.0, 0.5, 1.0]))
        thresh = float(rng.choice([1.0, 2.0, 3.5]))
        root_sift = int(rng.integers(0, 4) == 0)
        lowest = float(rng.choice([0.0, 0.0, 0.0, 2.0, 4.0]))
        subs = float(rng.choice([1.0, 1.0, 1.0, 2.0]))
        virtual = case % 3 == 2
        kind = int(rng.integers(0, 3))
        if kind == 0:
            img = synth.tile(int(rng.integers(0, 10 ** 6)), W, H, blur)
        elif kind == 1:
            img = synth.blobs(int(rng.integers(0, 10 ** 6)), W, H)
        else:
            img = rng.integers(0, 256, size=(H, W)).astype(np.float32)
        img = np.ascontiguousarray(img, dtype=np.float32)
        prm = capi.default_params(num_octaves=n_oct, init_blur=blur, peak_thresh=thresh, max_pts=1 << 17,
                                  root_sift=root_sift, lowest_scale=lowest, subsampling=subs)
        tag = "case %d: %dx%d P=%d oct=%d halo=%d blur=%.1f thr=%.1f kind=%d root=%d low=%g sub=%g %s" % (
            case, W, H, P, n_oct, halo, blur, thresh, kind, root_sift, lowest, subs, "virtual" if virtual else "threads")
        d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
        h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
        n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
        d_pts.free()
        if n >= prm.max_pts:
            print("skip %s (whole image saturates max_pts)" % tag, flush=True)
            continue
        want = canonical_order(h_pts[:n])
        full = torch.from_numpy(img).to(dev)
        torch.cuda.synchronize()
        try:
            if virtual:
                exts = [StripExtractor(k, P, W, H, prm, device=dev, halo=halo, strict=False) for k in range(P)]
                b = exts[0].plan.bounds
                parts = run_virtual(exts, [full[b[k]:b[k + 1]] for k in range(P)])
                flags = sum(e.check() for e in exts)
                for e in exts:
                    e.close()
                merged = None
            else:
                bounds = [(k * H) // P for k in range(P + 1)]

                def rank_fn(rank, make_comm):
                    with torch.cuda.device(dev):
                        c = capi.Context(0)
                        comm = make_comm(c)
                        ext = StripExtractor(rank, P, W, H, prm, device=dev, halo=halo, comm=comm, strict=False)
                        pts, cnt = run_distributed(ext, full[bounds[rank]:bounds[rank + 1]])
                        f = ext.check()
                        mine = ext.result()
                        g = SiftGatherer(comm, 1, prm.max_pts, region_cap=prm.max_pts, device=dev)
                        counts, gathered, totals = g.gather(pts, cnt)
                        c.synchronize()
                        mg = np.concatenate([x.cpu().numpy() for x in SiftGatherer.regions(gathered, totals)])
                        g.close()
                        ext.close()
                        comm.close()
                        c.close()
                        return mine, f, mg.view(SIFT_POINT_DTYPE).reshape(-1)

                res, errors = run_threads(P, rank_fn)
                if errors:
                    raise errors[0][1]
                parts = [r[0] for r in res]
                flags = sum(r[1] for r in res)
                merged = [r[2] for r in res]
            ok = True
            why = ""
            for pts in parts:
                if len(pts) and not np.all(np.diff(pts["subsampling"]) <= 0):
                    ok, why = False, "octave order"
            if flags:
                flagged += 1
                print("flag %s: %d keypoint(s) beyond the halo" % (tag, flags), flush=True)
                continue
            got = canonical_order(np.concatenate(parts)) if parts else want[:0]
            if ok and not same(got, want):
                ok, why = False, "union != whole image (%d vs %d)" % (len(got), len(want))
            if ok and merged is not None:
                for k, mg in enumerate(merged):
                    if not same(canonical_order(mg), want):
                        ok, why = False, "rank %d's merged SiftData differs" % k
                        break
            if ok:
                print("ok   %s keypoints=%d" % (tag, n), flush=True)
            else:
                failed += 1
                print("FAIL %s: %s" % (tag, why), flush=True)
        except Exception as e:  # noqa: BLE001
            failed += 1
            print("FAIL %s: %s: %s" % (tag, type(e).__name__, e), flush=True)
    st = fake_stats()
    print("%d of %d cases failed, %d flagged a shallow halo; transport: %d groups, %d sends, %.1f MB, %d size mismatches, "
          "%d timeouts" % (failed, n_cases, flagged, st["groups"], st["sends"], st["bytes"] / 1e6, st["mismatches"],
                           st["timeouts"]), flush=True)
    ctx.close()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
