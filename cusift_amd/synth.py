"""Synthetic inputs for the benchmark and the full-size tests.

`tile(seed)`  : mirror-tile the 640x480 fixture image to cover the target size, apply the gain
                255/144 (the fixture is dark: max 144), cyclic-shift by a splitmix64(seed)-derived
                offset and round to integers -- an 8-bit-valued float image with natural statistics.
`blobs(seed)` : 128 + Gaussian blobs + uniform noise, clamped and rounded.
Both are deterministic functions of (seed, w, h); no network, no dataset.
"""
import os

import numpy as np

_MASK = (1 << 64) - 1
_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gray1.pgm")


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & _MASK
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return state, z ^ (z >> 31)


def read_pgm(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"P5":
            raise ValueError("not a binary PGM: %s" % path)
        w, h = map(int, f.readline().split())
        maxval = int(f.readline())
        if maxval != 255:
            raise ValueError("only 8-bit PGM supported")
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w).astype(np.float32)


def write_pgm(path, img):
    a = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


_fixture_cache = None


def fixture_image():
    global _fixture_cache
    if _fixture_cache is None:
        _fixture_cache = read_pgm(_FIXTURE)
    return _fixture_cache


def gaussian_blur(img, sigma):
    """Separable Gaussian (radius ceil(4 sigma), replicated borders) in float32 -- the caller-side
    pre-blur of main.cpp:308-309 (cv::GaussianBlur) that `initBlur` describes."""
    if sigma <= 0:
        return img.astype(np.float32)
    r = int(np.ceil(4 * sigma))
    k = np.exp(-np.arange(-r, r + 1, dtype=np.float64) ** 2 / (2.0 * sigma * sigma))
    k = (k / k.sum()).astype(np.float32)
    a = np.pad(img.astype(np.float32), ((r, r), (r, r)), mode="edge")
    h, w = img.shape
    tmp = np.zeros((h + 2 * r, w), dtype=np.float32)
    for i in range(2 * r + 1):
        tmp += k[i] * a[:, i:i + w]
    out = np.zeros((h, w), dtype=np.float32)
    for i in range(2 * r + 1):
        out += k[i] * tmp[i:i + h, :]
    return out


_tiled_cache = {}


def _mirror_tiled(ny, nx):
    if (ny, nx) not in _tiled_cache:
        g = fixture_image()
        row = [g if i % 2 == 0 else g[:, ::-1] for i in range(nx)]
        strip = np.concatenate(row, axis=1)
        rows = [strip if j % 2 == 0 else strip[::-1, :] for j in range(ny)]
        _tiled_cache[(ny, nx)] = np.ascontiguousarray(np.concatenate(rows, axis=0))
    return _tiled_cache[(ny, nx)]


def _tile_plain(seed, w, h, preblur):
    g = fixture_image()
    gh, gw = g.shape
    ny, nx = -(-h // gh) + 1, -(-w // gw) + 1
    big = _mirror_tiled(ny, nx)
    st, r0 = splitmix64(seed)
    st, r1 = splitmix64(st)
    # cyclic shift of the mirror-tiled plane == modular row/column gather
    rows = (np.arange(h) - int(r1 % gh)) % big.shape[0]
    cols = (np.arange(w) - int(r0 % gw)) % big.shape[1]
    big = big[rows][:, cols]
    out = big[:h, :w] * np.float32(255.0 / 144.0)
    if preblur > 0:
        out = gaussian_blur(out, preblur)
    return np.clip(np.rint(out), 0, 255).astype(np.float32)


def tile(seed, w=1920, h=1080, preblur=0.0):
    """preblur > 0: the image is low-passed to that sigma and re-quantised to 8 bit, so that it really
    carries the blur a caller declares with initBlur (BASELINE config: initBlur = 1.0).

    Large blurred images (BASELINE configs[4]: 8192 x 8192) are assembled from ONE period of the pattern: away from its
    border -- and from the rows / columns where the cyclic shift wraps around a plane with an odd number of tiles -- the
    mirror-tiled image is periodic with period (2 gh, 2 gw), so is its blur, and every pixel's sum is formed from the same
    values in the same order: bit for bit _tile_plain's result (tests/test_frontend.py) in 2 s instead of 20."""
    g = fixture_image()
    gh, gw = g.shape
    ph, pw = 2 * gh, 2 * gw
    if preblur <= 0 or h < 4 * ph or w < 4 * pw:
        return _tile_plain(seed, w, h, preblur)
    r = int(np.ceil(4 * preblur))
    ny, nx = -(-h // gh) + 1, -(-w // gw) + 1
    big = _mirror_tiled(ny, nx)
    st, r0 = splitmix64(seed)
    st, r1 = splitmix64(st)
    sy, sx = int(r1 % gh), int(r0 % gw)
    rows = (np.arange(h) - sy) % big.shape[0]
    cols = (np.arange(w) - sx) % big.shape[1]
    scale = np.float32(255.0 / 144.0)

    def raw(y0, y1, x0, x1):
        return big[rows[y0:y1]][:, cols[x0:x1]] * scale

    # frame: r pixels at every border (replicated edges) plus, at the top / left, the rows / columns that come from the
    # far end of a plane with an odd number of tiles (there the wrap-around breaks the mirror pattern)
    mt = r + (sy if ny % 2 else 0)
    ml = r + (sx if nx % 2 else 0)
    out = np.empty((h, w), dtype=np.float32)
    core = gaussian_blur_valid(raw(mt - r, mt + ph + r, ml - r, ml + pw + r), preblur)  # pixels [mt, mt+ph) x [ml, ml+pw)
    reps_y, reps_x = -(-(h - mt) // ph), -(-(w - ml) // pw)
    out[mt:, ml:] = np.tile(core, (reps_y, reps_x))[:h - mt, :w - ml]
    # the frame, with the plain blur on strips that carry r true neighbours beyond what is kept
    out[:mt, :] = gaussian_blur(raw(0, mt + r, 0, w), preblur)[:mt, :]
    out[h - r:, :] = gaussian_blur(raw(h - 2 * r, h, 0, w), preblur)[r:, :]
    out[:, :ml] = gaussian_blur(raw(0, h, 0, ml + r), preblur)[:, :ml]
    out[:, w - r:] = gaussian_blur(raw(0, h, w - 2 * r, w), preblur)[:, r:]
    return np.clip(np.rint(out), 0, 255).astype(np.float32)


def gaussian_blur_valid(img, sigma):
    """gaussian_blur's sums without the padding: output pixel (y, x) is the blur of img centred at (y + r, x + r), formed
    from the same taps in the same order; shape (H - 2 r, W - 2 r)."""
    r = int(np.ceil(4 * sigma))
    k = np.exp(-np.arange(-r, r + 1, dtype=np.float64) ** 2 / (2.0 * sigma * sigma))
    k = (k / k.sum()).astype(np.float32)
    a = img.astype(np.float32)
    h, w = a.shape[0] - 2 * r, a.shape[1] - 2 * r
    tmp = np.zeros((a.shape[0], w), dtype=np.float32)
    for i in range(2 * r + 1):
        tmp += k[i] * a[:, i:i + w]
    out = np.zeros((h, w), dtype=np.float32)
    for i in range(2 * r + 1):
        out += k[i] * tmp[i:i + h, :]
    return out


def blobs(seed, w=1920, h=1080):
    rng = np.random.default_rng(seed)
    n = max(1, w * h // 400)
    img = np.full((h, w), 128.0, dtype=np.float32)
    cx = rng.uniform(0, w, n)
    cy = rng.uniform(0, h, n)
    sg = np.exp(rng.uniform(np.log(1.0), np.log(24.0), n))
    am = rng.uniform(-64, 64, n)
    for i in range(n):
        r = int(np.ceil(3 * sg[i]))
        x0, x1 = max(0, int(cx[i]) - r), min(w, int(cx[i]) + r + 1)
        y0, y1 = max(0, int(cy[i]) - r), min(h, int(cy[i]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1, dtype=np.float32) - cx[i]
        ys = np.arange(y0, y1, dtype=np.float32) - cy[i]
        img[y0:y1, x0:x1] += (am[i] * np.exp(-(ys[:, None] ** 2 + xs[None, :] ** 2) / (2 * sg[i] ** 2))).astype(
            np.float32)
    img += rng.uniform(-4, 4, (h, w)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.float32)


def batch(n, w=1920, h=1080, first_seed=1000, preblur=0.0):
    """n tile images, seeds first_seed .. first_seed+n-1, as one (n, h, w) float32 array."""
    return np.stack([tile(first_seed + i, w, h, preblur) for i in range(n)], axis=0)
