"""Coefficients of cusift_amd/csrc/sift_math.h: weighted least squares with Lawson iteration (near-minimax in relative
error), rounded to float32.  Run: python tools/fit_math_polys.py -- the accuracy actually achieved by the C code
is measured by tests/test_math.py."""
import numpy as np
from numpy.polynomial import polynomial as P

def remez_like(f, lo, hi, deg, weight, iters=40, n=20001):
    """weighted least squares with Lawson iteration -> near-minimax; returns monomial coeffs (low->high)"""
    x = 0.5*(lo+hi) + 0.5*(hi-lo)*np.cos(np.pi*(np.arange(n)+0.5)/n)
    y = f(x); w = weight(x)
    lw = np.ones_like(x)
    best=None
    for it in range(iters):
        A = np.vander(x, deg+1, increasing=True) * (w*np.sqrt(lw))[:,None]
        c, *_ = np.linalg.lstsq(A, y*w*np.sqrt(lw), rcond=None)
        err = np.abs((np.vander(x, deg+1, increasing=True)@c - y)*w)
        if best is None or err.max() < best[0]: best=(err.max(), c.copy())
        lw = lw*(err/err.mean()+1e-30); lw/=lw.mean()
    return best

# atan(a) = a + a*s*Q(s), s=a^2 in [0,1]; Q(s) = (atan(a)/a - 1)/s
def fQ(s):
    a=np.sqrt(s); 
    with np.errstate(all='ignore'):
        q = (np.arctan(a)/a - 1)/s
    return q
# relative error of atan ~ a*s*dQ / atan(a)
for deg in (7,8):
    e,c = remez_like(fQ, 1e-12, 1.0, deg, lambda s: np.sqrt(s)*s/np.arctan(np.sqrt(s)))
    print("atan Q deg",deg,"max rel err", e, "ulp~", e/2**-24)
    print([float(np.float32(v)) for v in c])

def fP(r):
    with np.errstate(all='ignore'):
        return (np.expm1(r) - r)/(r*r)
for deg in (4,5):
    e,c = remez_like(fP, -0.3467, 0.3467, deg, lambda r: r*r/np.exp(r))
    print("exp P deg",deg,"ulp~", e/2**-24); print([float(np.float32(v)) for v in c])
# 2^r = 1 + r*G(r), |r|<=0.5
def fG(r):
    with np.errstate(all='ignore'):
        return np.expm1(r*np.log(2))/r
for deg in (5,6):
    e,c = remez_like(fG, -0.5, 0.5, deg, lambda r: np.abs(r)/np.exp2(r))
    print("exp2 G deg",deg,"ulp~", e/2**-24); print([float(np.float32(v)) for v in c])
import math
pio2=math.pi/2
print("pio2 hi", repr(float(np.float64(pio2))), "lo", repr(np.float64(np.longdouble(np.pi)/2 - np.longdouble(np.float64(pio2)))))
print("PI f hi", repr(float(np.float32(math.pi))), "lo", repr(float(np.float32(math.pi-float(np.float32(math.pi))))))
print("PIO2 f hi", repr(float(np.float32(pio2))), "lo", repr(float(np.float32(pio2-float(np.float32(pio2))))))
print("log2e f", repr(float(np.float32(1/math.log(2)))))
ln2=math.log(2); hi=float(np.float32(ln2)); 
import struct
b=struct.unpack('<I',struct.pack('<f',ln2))[0] & 0xfffff000
hi=struct.unpack('<f',struct.pack('<I',b))[0]
print("ln2 hi",repr(hi),"lo",repr(float(np.float32(ln2-hi))))
