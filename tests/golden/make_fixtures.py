#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/ from the reference's own test data.

Run in the build container only (the reference tree does not exist on the GPU box):

    python tests/golden/make_fixtures.py [/root/reference]

Fixtures are DATA the reference's tests hold (test/detector.cpp:52-84 writes/reads them):

* gray1.pgm        <- test/data/gray1      640x480 float32 raw, integer valued 0..144 (libjpeg Y
                                           channel of color1.jpg).  Stored as a binary P5 PGM
                                           (lossless because every value is an integer < 256).
* cusift1_check.bin<- test/data/cusift1_check   u32 n=4096, then n x {x, y, scale, orientation} f32
* cusift1.bin      <- test/data/cusift1         a second run of the reference, same format

Parameters that produced the two keypoint files (test/detector.cpp:37-49):
    numOctaves=6 initBlur=0.0 peakThresh=0.1 edgeThresh=10 lowestScale=0 subsampling=1 maxPts=4096
"""
import os
import sys
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(ref, "test", "data")

g = np.fromfile(os.path.join(src, "gray1"), dtype="<f4")
assert g.size == 640 * 480, g.size
assert np.all(g == np.round(g)) and g.min() >= 0 and g.max() <= 255, (g.min(), g.max())
with open(os.path.join(here, "gray1.pgm"), "wb") as f:
    f.write(b"P5\n640 480\n255\n")
    f.write(g.astype(np.uint8).tobytes())

for name in ("cusift1_check", "cusift1"):
    raw = open(os.path.join(src, name), "rb").read()
    n = int(np.frombuffer(raw[:4], dtype="<u4")[0])
    assert len(raw) == 4 + n * 16, (name, len(raw), n)
    with open(os.path.join(here, name + ".bin"), "wb") as f:
        f.write(raw)
    print(name, "numPts =", n)
print("gray1: min/max", g.min(), g.max())

# ---- matcher fixtures (test/test.cpp:25-56) -------------------------------
# sift/sift1, sift/sift2 : VLFeat dumps  u32 n; f32[n][4] (x, y, scale, orientation); f32[n][128] descriptors
#                          (extras/debug.cpp:118-165)
# match_indices1_2       : MATLAB match indices  u32 n; u32 i[n]; u32 j[n], 1-based (extras/debug.cpp:167-181)
# The reference's MatchingRatioTest expects 340 matches for MatchSiftData(L2, 1000, 0.6) on this pair.
import shutil
for rel, name in (("sift/sift1", "vlfeat_sift1.bin"), ("sift/sift2", "vlfeat_sift2.bin"),
                  ("match_indices/match_indices1_2", "match_indices1_2.bin")):
    shutil.copyfile(os.path.join(src, rel), os.path.join(here, name))
    os.chmod(os.path.join(here, name), 0o644)
    print(name, os.path.getsize(os.path.join(here, name)), "bytes")
