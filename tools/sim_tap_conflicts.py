"""Bank-conflict degree of the descriptor taps (64 lanes reading the 2x2 footprints of a rotated 16x4 sample lattice) for
different patch row strides / skews; CPU only.  Stride 40 (the shipped layout) is the best of those tried: 2.23."""
import numpy as np
rng = np.random.default_rng(1)
def degree(addr):
    # addr: [64] dword addresses; conflict degree = max over banks of #distinct addresses
    banks = addr % 64
    deg = 0
    for b in np.unique(banks):
        deg = max(deg, len(np.unique(addr[banks == b])))
    return deg
def addr_fn(kind):
    if kind == "s40": return lambda r, c: 40 * r + c
    if kind == "s41": return lambda r, c: 41 * r + c
    if kind == "s40skew1": return lambda r, c: 40 * r + c + (r >> 3)
    if kind == "s40skew_r": return lambda r, c: 40 * r + c + (r & 7) * 0 + (r >> 3) * 3
    if kind == "s42": return lambda r, c: 42 * r + c
    if kind == "s44": return lambda r, c: 44 * r + c
    if kind == "s48": return lambda r, c: 48 * r + c
kinds = ["s40", "s41", "s40skew1", "s40skew_r", "s42", "s44"]
tot = {k: 0.0 for k in kinds}
n = 0
lane = np.arange(64)
tx = lane % 16
for it in range(400):
    theta = rng.uniform(0, 2 * np.pi)
    scale = rng.uniform(1.0, 2.0)
    s = 0.75 * scale
    px, py = rng.uniform(20, 21, 2)
    ca, sa = np.cos(theta), np.sin(theta)
    for step in range(4):
        y = lane // 16 + 4 * step
        xpos = px + (tx - 7.5) * s * ca - (y - 7.5) * s * sa
        ypos = py + (tx - 7.5) * s * sa + (y - 7.5) * s * ca
        for (ox, oy) in ((ca, sa), (-ca, -sa), (-sa, ca), (sa, -ca)):
            c = np.floor(xpos + ox - 0.5).astype(int)
            r = np.floor(ypos + oy - 0.5).astype(int)
            r -= r.min(); c -= c.min()
            for k in kinds:
                f = addr_fn(k)
                # a tap = read2 (c, c+1) on row r and on row r+1: four dword accesses
                d = 0
                for (dr, dc) in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    d += degree(f(r + dr, c + dc))
                tot[k] += d
            n += 4
for k in kinds:
    print(k, "mean conflict degree per dword access: %.3f" % (tot[k] / n))
