import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from cusift_amd import synth
from cusift_amd.batch import PipelinedExtractor
B, w, h = 64, 1920, 1080
pipe = PipelinedExtractor(B, w, h, n_streams=4, num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0,
                          lowest_scale=0.0, subsampling=1.0, max_pts=32768, tex_frac_bits=8)
imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(B)])
d = pipe.extractors[0].images_from_numpy(imgs)
for _ in range(8): pipe.submit(d)
pipe.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): pipe.submit(d)
t1 = time.perf_counter()
pipe.synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms per step (host), total %.3f ms per step" % ((t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3))
