"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the oracle is the arbiter of
every parity claim, so its own memory safety is checked).  `make -C oracle asan` builds libsift_oracle_asan.so; a child
process with libasan preloaded runs the golden fixture and randomised inputs (odd sizes, tiny images, saturation,
non-zero initBlur, lowestScale, both texture-fraction models, the matcher and the homography) through it and must
produce the same results as the plain build with no sanitizer report.

CPU only; GPU sanitizers are not available on this pool.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(tests)r)
import numpy as np
from conftest import read_pgm
from oracle_binding import Oracle, pitched, SIFT_POINT_DTYPE

san, ref = Oracle("asan"), Oracle("")
g = read_pgm(%(gray)r)
kw = dict(num_octaves=6, init_blur=0.0, peak_thresh=0.1, edge_thresh=10.0, max_pts=16384)
a, b = san.extract(g, **kw), ref.extract(g, **kw)
assert len(a) == len(b) == 9508 and a.tobytes() == b.tobytes()
a = san.extract(g, num_octaves=6, peak_thresh=0.1, max_pts=100)   # saturation: the append guard
assert len(a) == 100
rng = np.random.default_rng(11)
for case in range(40):
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    img = np.clip(np.rint(128 + 50 * rng.standard_normal((h, w))), 0, 255).astype(np.float32)
    if case %% 5 == 0:
        img[:] = 77.0   # flat: NaN orientations / descriptors
    prm = dict(num_octaves=int(rng.integers(1, 8)), init_blur=float(rng.choice([0.0, 0.5, 1.0, 1.3])),
               peak_thresh=float(rng.choice([0.5, 2.0, 8.0])), edge_thresh=float(rng.choice([5.0, 10.0])),
               lowest_scale=float(rng.choice([0.0, 0.0, 2.0])), subsampling=float(rng.choice([1.0, 2.0])),
               max_pts=int(rng.choice([16, 256, 4096])), tex_frac_bits=int(rng.choice([0, 8])))
    x, y = san.extract(img, **prm), ref.extract(img, **prm)
    assert len(x) == len(y) and x.tobytes() == y.tobytes(), (case, w, h, prm)
    if w >= 2 and h >= 2:
        d = san.scale_down(pitched(img), w, h)
    dog = san.laplace_multi(pitched(img), w, h, prm["init_blur"])
    san.find_points_multi(dog, w, h, prm["peak_thresh"], 10.0, 1.0, 8)    # tiny capacity
# matcher + homography on extracted points
p1 = san.extract(g, num_octaves=4, peak_thresh=1.0, max_pts=4096)
p2 = san.extract(np.roll(g, (5, 9), axis=(0, 1)), num_octaves=4, peak_thresh=1.0, max_pts=4096)
for dist in (0, 1):
    san.match(p1, p2, dist)
    san.match(p1[:7], p2[:3], dist)
idx = san.match_filter(p1, 1000.0, 0.9)
rp = rng.integers(0, len(p1), (4, 64)).astype(np.int32)
san.find_homography(p1, rp, 5.0)
san.rootsift(p1, len(p1))
x = np.linspace(-200, 200, 4001).astype(np.float32)
for op in ("exp", "exp2", "sincos"):
    san.math_eval(op, x)
san.math_eval("atan2", x, x[::-1].copy())
print("SANITIZED-OK")
'''


def test_oracle_runs_clean_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan is not installed")
    rc = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr[-2000:]
    env = dict(os.environ)
    env["LD_PRELOAD"] = os.path.realpath(libasan)
    # python itself leaks by design; everything else is fatal
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    code = CHILD % {"tests": os.path.join(ROOT, "tests"), "gray": os.path.join(ROOT, "tests", "golden", "gray1.pgm")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    tail = (out.stdout + out.stderr)[-4000:]
    assert out.returncode == 0 and "SANITIZED-OK" in out.stdout, tail
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, tail
