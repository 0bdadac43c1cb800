"""ctypes binding of oracle/libsift_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# cuSIFT.h:10-30 -- 588-byte record
SIFT_POINT_DTYPE = np.dtype(
    [
        ("coords2D", "<f4", (2,)),
        ("scale", "<f4"),
        ("sharpness", "<f4"),
        ("edgeness", "<f4"),
        ("orientation", "<f4"),
        ("score", "<f4"),
        ("ambiguity", "<f4"),
        ("match", "<i4"),
        ("match_xpos", "<f4"),
        ("match_ypos", "<f4"),
        ("match_error", "<f4"),
        ("subsampling", "<f4"),
        ("empty", "<f4", (3,)),
        ("data", "<f4", (128,)),
        ("coords3D", "<f4", (3,)),
    ]
)
assert SIFT_POINT_DTYPE.itemsize == 588


class OracleParams(C.Structure):
    _fields_ = [
        ("num_octaves", C.c_int),
        ("init_blur", C.c_double),
        ("peak_thresh", C.c_float),
        ("edge_thresh", C.c_float),
        ("lowest_scale", C.c_float),
        ("subsampling", C.c_float),
        ("max_pts", C.c_int),
        ("tex_frac_bits", C.c_int),
    ]


def build_oracle():
    """Compile the oracle if the shared objects are missing or stale (needs gcc)."""
    so = os.path.join(ORACLE_DIR, "libsift_oracle.so")
    deps = [os.path.join(ORACLE_DIR, "sift_oracle.c"), os.path.join(ORACLE_DIR, "sift_oracle.h"),
            os.path.join(ROOT, "cusift_amd", "csrc", "sift_math.h")]
    sos = [so, os.path.join(ORACLE_DIR, "libsift_oracle_libm.so"), os.path.join(ORACLE_DIR, "libsift_oracle_nofma.so")]
    stale = any(not os.path.exists(x) for x in sos) or any(
        os.path.exists(d) and os.path.getmtime(d) > min(os.path.getmtime(x) for x in sos) for d in deps)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return so


_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class Oracle:
    def __init__(self, variant=""):
        build_oracle()
        name = "libsift_oracle%s.so" % (("_" + variant) if variant else "")
        self.lib = lib = C.CDLL(os.path.join(ORACLE_DIR, name))
        lib.oracle_scale_down.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int]
        lib.oracle_scale_down.restype = None
        lib.oracle_laplace_taps.argtypes = [C.c_float, _vp]
        lib.oracle_laplace_taps.restype = None
        lib.oracle_laplace_multi.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp]
        lib.oracle_laplace_multi.restype = None
        lib.oracle_find_points_multi.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                                 _vp, C.c_int, C.POINTER(C.c_int)]
        lib.oracle_find_points_multi.restype = None
        lib.oracle_compute_orientations.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int]
        lib.oracle_compute_orientations.restype = None
        lib.oracle_extract_descriptors.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_float,
                                                   C.c_int]
        lib.oracle_extract_descriptors.restype = None
        lib.oracle_rootsift.argtypes = [_vp, C.c_int]
        lib.oracle_rootsift.restype = None
        lib.oracle_extract.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(OracleParams), _vp]
        lib.oracle_extract.restype = C.c_int
        lib.oracle_set_orientation_diag.argtypes = [_vp, C.c_int]
        lib.oracle_set_orientation_diag.restype = None
        lib.oracle_match_sift_data.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_int]
        lib.oracle_match_sift_data.restype = None
        lib.oracle_match_filter.argtypes = [_vp, C.c_int, C.c_float, C.c_float, _vp]
        lib.oracle_match_filter.restype = C.c_int
        lib.oracle_u8_to_f32.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int]
        lib.oracle_u8_to_f32.restype = None
        lib.oracle_gaussian3x3.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_float]
        lib.oracle_gaussian3x3.restype = None
        lib.oracle_tex2d.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        lib.oracle_tex2d.restype = C.c_float
        lib.oracle_find_homography.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_float, _vp, C.POINTER(C.c_int), _vp,
                                               _vp]
        lib.oracle_find_homography.restype = C.c_int

    # ---- stage functions on pitched numpy images (2-D float32 arrays, pitch = arr.shape[1]) ----
    def scale_down(self, src, w, h):
        src = _f32(src)
        ow, oh = w // 2, h // 2
        op = max(128, -(-ow // 128) * 128)
        dst = np.zeros((max(oh, 1), op), dtype=np.float32)
        self.lib.oracle_scale_down(src.ctypes.data, w, h, src.shape[1], dst.ctypes.data, op)
        return dst

    def laplace_taps(self, init_blur):
        taps = np.zeros(8 * 16, dtype=np.float32)
        self.lib.oracle_laplace_taps(init_blur, taps.ctypes.data)
        return taps

    def laplace_multi(self, img, w, h, init_blur):
        img = _f32(img)
        dog = np.zeros((7, h, img.shape[1]), dtype=np.float32)
        self.lib.oracle_laplace_multi(img.ctypes.data, w, h, img.shape[1], init_blur, dog.ctypes.data)
        return dog

    def find_points_multi(self, dog, w, h, peak_thresh, edge_thresh, subsampling, max_pts, points=None, counter=0):
        dog = _f32(dog)
        if points is None:
            points = np.zeros(max_pts, dtype=SIFT_POINT_DTYPE)
        cnt = C.c_int(counter)
        self.lib.oracle_find_points_multi(dog.ctypes.data, w, h, dog.shape[2], peak_thresh, edge_thresh, subsampling,
                                          points.ctypes.data, max_pts, C.byref(cnt))
        return points, cnt.value

    def compute_orientations(self, img, w, h, points, first, last, frac_bits=8):
        img = _f32(img)
        self.lib.oracle_compute_orientations(img.ctypes.data, w, h, img.shape[1], points.ctypes.data, first, last,
                                             frac_bits)

    def extract_descriptors(self, img, w, h, points, first, last, subsampling, frac_bits=8):
        img = _f32(img)
        self.lib.oracle_extract_descriptors(img.ctypes.data, w, h, img.shape[1], points.ctypes.data, first, last,
                                            subsampling, frac_bits)

    def rootsift(self, points, n):
        self.lib.oracle_rootsift(points.ctypes.data, n)

    def extract(self, img, num_octaves=5, init_blur=0.0, peak_thresh=3.0, edge_thresh=10.0, lowest_scale=0.0,
                subsampling=1.0, max_pts=32768, tex_frac_bits=8):
        img = _f32(img)
        h, w = img.shape
        prm = OracleParams(num_octaves, init_blur, peak_thresh, edge_thresh, lowest_scale, subsampling, max_pts,
                           tex_frac_bits)
        points = np.zeros(max_pts, dtype=SIFT_POINT_DTYPE)
        n = self.lib.oracle_extract(img.ctypes.data, w, h, C.byref(prm), points.ctypes.data)
        return points[:n]

    def extract_with_orientation_peaks(self, img, **kw):
        """extract() plus, per returned point, (second / first smoothed-histogram peak, the orientation the second peak
        would give in degrees) -- the diagnostic tap of oracle_compute_orientations (cuSIFT_D.cu:362-394 computes both
        peaks and keeps the first)."""
        max_pts = kw.get("max_pts", 32768)
        diag = np.full((max_pts, 2), np.nan, dtype=np.float32)
        self.lib.oracle_set_orientation_diag(diag.ctypes.data, max_pts)
        try:
            pts = self.extract(img, **kw)
        finally:
            self.lib.oracle_set_orientation_diag(None, 0)
        return pts, diag[:len(pts)]

    def match(self, sift1, sift2, distance=1):
        """MatchSiftData: fills score/ambiguity/match/match_xpos/match_ypos of sift1 in place."""
        self.lib.oracle_match_sift_data(sift1.ctypes.data, len(sift1), sift2.ctypes.data, len(sift2), distance)

    def match_filter(self, sift1, score_threshold=999.0, ambiguity_threshold=1.0):
        idx = np.zeros(max(len(sift1), 1), dtype=np.int32)
        n = self.lib.oracle_match_filter(sift1.ctypes.data, len(sift1), score_threshold, ambiguity_threshold,
                                         idx.ctypes.data)
        return idx[:n]

    def u8_to_f32(self, img_u8):
        img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
        h, w = img_u8.shape
        out = np.zeros((h, w), dtype=np.float32)
        self.lib.oracle_u8_to_f32(img_u8.ctypes.data, w, h, w, out.ctypes.data, w)
        return out

    def gaussian3x3(self, img, sigma):
        img = _f32(img)
        h, w = img.shape
        out = np.zeros((h, w), dtype=np.float32)
        self.lib.oracle_gaussian3x3(img.ctypes.data, w, h, w, out.ctypes.data, w, sigma)
        return out

    def find_homography(self, points, rand_pts, thresh=5.0):
        """Device part + selection of FindHomography for the given samples (int32 [4, num_loops]).
        Returns (H[9], num_matches, best index, all homographies [8, L], all counts [L])."""
        rand_pts = np.ascontiguousarray(rand_pts, dtype=np.int32)
        loops = rand_pts.shape[1]
        hom = np.zeros(9, dtype=np.float32)
        n = C.c_int(0)
        all_h = np.zeros((8, loops), dtype=np.float32)
        all_c = np.zeros(loops, dtype=np.int32)
        best = self.lib.oracle_find_homography(points.ctypes.data, len(points), rand_pts.ctypes.data, loops, thresh,
                                               hom.ctypes.data, C.byref(n), all_h.ctypes.data, all_c.ctypes.data)
        return hom, n.value, best, all_h, all_c

    def math_eval(self, op, a, b=None):
        """The written-out transcendental functions shared with the kernels (cusift_amd/csrc/sift_math.h), array form:
        op 'exp' | 'exp2' | 'atan2' (a = y, b = x) | 'sincos' (returns (sin, cos))."""
        code = {"exp": 0, "exp2": 1, "atan2": 2, "sincos": 3}[op]
        a = _f32(a).ravel()
        b = _f32(b).ravel() if b is not None else a
        out, out2 = np.zeros_like(a), np.zeros_like(a)
        self.lib.oracle_math_eval.argtypes = [C.c_int, _vp, _vp, _vp, _vp, C.c_int]
        self.lib.oracle_math_eval.restype = None
        self.lib.oracle_math_eval(code, a.ctypes.data, b.ctypes.data, out.ctypes.data, out2.ctypes.data, a.size)
        return (out, out2) if op == "sincos" else out

    def tex2d(self, img, w, h, x, y, frac_bits=8):
        img = _f32(img)
        return self.lib.oracle_tex2d(img.ctypes.data, w, h, img.shape[1], x, y, frac_bits)


def pitched(img):
    """Dense (h, w) -> pitched (h, iAlignUp(w,128)) copy, pad columns zero (cuImage.cu:11-13)."""
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape
    p = -(-w // 128) * 128
    out = np.zeros((h, p), dtype=np.float32)
    out[:, :w] = img
    return out


def read_vlfeat_sift(path):
    """VLFeat dump (extras/debug.cpp:118-165): u32 n; f32[n][4] x,y,scale,orientation; f32[n][128] -> SiftPoint array."""
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], dtype="<u4")[0])
    pts = np.frombuffer(raw[4:4 + 16 * n], dtype="<f4").reshape(n, 4)
    desc = np.frombuffer(raw[4 + 16 * n:], dtype="<f4").reshape(n, 128)
    out = np.zeros(n, dtype=SIFT_POINT_DTYPE)
    out["coords2D"] = pts[:, :2]
    out["scale"] = pts[:, 2]
    out["orientation"] = pts[:, 3]
    out["data"] = desc
    return out


def read_match_indices(path):
    """MATLAB match indices (extras/debug.cpp:167-181): u32 n; u32 i[n]; u32 j[n], 1-based."""
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], dtype="<u4")[0])
    a = np.frombuffer(raw[4:], dtype="<u4")
    return a[:n].astype(np.int64), a[n:2 * n].astype(np.int64)
