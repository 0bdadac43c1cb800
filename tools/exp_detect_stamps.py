"""Where a wave of detect_fused_kernel spends its cycles: needs a library built with -DCUSIFT_DET_STAMPS
(tools/exp_detect_stamps.sh builds one and runs this).  Octave 0 of the bench's 64 x 1080p batch by default."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from cusift_amd import capi, synth
from cusift_amd.batch import BatchExtractor

B, w, h = 64, 1920, 1080
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ex = BatchExtractor(B, w, h, num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, lowest_scale=0.0,
                    subsampling=1.0, max_pts=4096, tex_frac_bits=8)
ex.params.concurrent_batches = conc
imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(B)])
d = ex.images_from_numpy(imgs)
L = capi.lib()
L.cusift_debug_det_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
names = ["window fill (9 rows)", "blur + DoG", "pre-test + extrema analysis", "refinement + key list",
         "window shift (+ wait for the row)", "wave total"]


def report(title):
    L.cusift_debug_det_cycles(out, 1)
    v = [int(x) for x in out]
    rows, waves, tot = v[6], v[7], v[5]
    print("%s: %d waves, %d wave-rows (%.1f per wave)" % (title, waves, rows, rows / waves))
    if os.environ.get("DET_STAMPS_MODE") == "3":
        ev = v[1]
        print("  rows with a candidate: %d of %d (%.1f%%); refinement segment: %.0f cycles on such a row, %.0f on the others"
              % (ev, rows, 100.0 * ev / rows, v[2] / max(ev, 1), v[3] / max(rows - ev, 1)))
        return
    if os.environ.get("DET_STAMPS_MODE") == "4":
        ev = max(v[1], 1)
        print("  scale-events (a row x scale with a candidate): %d (%.3f per wave-row); cube dump %.0f cycles each, copy into "
              "the list %.0f, batch refinement %.0f per batch (%d batches inside the loop)"
              % (v[1], v[1] / rows, v[2] / ev, v[3] / ev, v[0] / max(v[4], 1), v[4]))
        return
    for n, c in zip(names, v[:6]):
        print("  %-36s %8.0f cycles/wave %8.1f cycles/wave-row %6.1f%%" % (n, c / waves, c / rows, 100.0 * c / tot))
    print("  %-36s %8.0f cycles/wave" % ("unattributed (stamps, flush, loop)", (tot - sum(v[:5])) / waves))


ex.extract(d)
torch.cuda.synchronize()
L.cusift_debug_det_cycles(out, 1)
# octave 0 alone: the stage entry point on the batch's images (chunk height as for one stream)
pts = torch.zeros((B, 4096, 147), dtype=torch.float32, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
for rep in range(2):
    cnt.zero_()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    L.cusift_debug_det_cycles(out, 1)
    t0.record(torch.cuda.current_stream())
    ex.ctx.detect_multi(d.data_ptr(), w, h, ex.pitch, 1.0, 3.0, 10.0, 1.0, pts.data_ptr(), 4096, cnt.data_ptr(), B,
                        img_stride=h * ex.pitch)
    t1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
print("octave 0 launch: %.3f ms, %d candidates kept" % (t0.elapsed_time(t1), int(cnt.sum())))
report("octave 0 of 64 x 1080p (cusift_detect_multi)")
