"""ctypes binding of the C ABI declared in include/cusift_amd.h (libcusift_amd.so).

There is no CPU fallback: if the HIP extension is missing this module raises at import of the
library handle (`lib()`), and every call that needs a GPU fails with the library's error text.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUSIFT_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcusift_amd.so")

CUSIFT_OK = 0
NUM_STAGES = 8
STAGE_NAMES = ("scale_down", "laplace_multi", "find_points_multi", "compute_orientations",
               "extract_descriptors", "total", "detect_multi", "describe_all")

# SiftPoint, cuSIFT.h:10-30 (588 B, no padding)
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
SIFT_POINT_BYTES = 588
assert SIFT_POINT_DTYPE.itemsize == SIFT_POINT_BYTES

# cusift_compact_point (include/cusift_amd.h): the optional 160-byte wire record
COMPACT_POINT_DTYPE = np.dtype([("coords2D", "<f4", (2,)), ("scale", "<f4"), ("sharpness", "<f4"), ("edgeness", "<f4"),
                                ("orientation", "<f4"), ("subsampling", "<f4"), ("desc_step", "<f4"), ("q", "u1", (128,))])
COMPACT_POINT_BYTES = 160
assert COMPACT_POINT_DTYPE.itemsize == COMPACT_POINT_BYTES

# cusift_trimmed_point: the 135 floats extraction writes, exact (optional 540-byte wire record)
TRIMMED_POINT_DTYPE = np.dtype([("coords2D", "<f4", (2,)), ("scale", "<f4"), ("sharpness", "<f4"), ("edgeness", "<f4"),
                                ("orientation", "<f4"), ("subsampling", "<f4"), ("data", "<f4", (128,))])
TRIMMED_POINT_BYTES = 540
assert TRIMMED_POINT_DTYPE.itemsize == TRIMMED_POINT_BYTES
WIRE_FORMATS = {"exact": (0, SIFT_POINT_BYTES), "compact": (1, COMPACT_POINT_BYTES), "trimmed": (2, TRIMMED_POINT_BYTES)}


class Params(C.Structure):
    """cusift_params (include/cusift_amd.h); the public parameter fields of SiftData, cuSIFT.h:44-51."""

    _fields_ = [
        ("num_octaves", C.c_int),
        ("init_blur", C.c_double),
        ("peak_thresh", C.c_float),
        ("edge_thresh", C.c_float),
        ("lowest_scale", C.c_float),
        ("subsampling", C.c_float),
        ("max_pts", C.c_int),
        ("tex_frac_bits", C.c_int),
        ("fused_detect", C.c_int),
        ("root_sift", C.c_int),
        ("concurrent_batches", C.c_int),
    ]


class CusiftError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/cusift_amd.h declares
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_PP = C.POINTER(Params)
SIGNATURES = {
    "cusift_last_error": (C.c_char_p, []),
    "cusift_version": (C.c_char_p, []),
    "cusift_device_count": (_i, [C.POINTER(_i)]),
    "cusift_init": (_i, [_i]),
    "cusift_default_params": (None, [_PP]),
    "cusift_ctx_create": (_i, [C.POINTER(_vp), _i, _vp]),
    "cusift_ctx_create_borrowed": (_i, [C.POINTER(_vp), _i, _vp]),
    "cusift_ctx_destroy": (_i, [_vp]),
    "cusift_ctx_synchronize": (_i, [_vp]),
    "cusift_ctx_stream": (_vp, [_vp]),
    "cusift_ctx_device": (_i, [_vp]),
    "cusift_ctx_wait": (_i, [_vp, _vp]),
    "cusift_comm_get_unique_id": (_i, [_vp]),
    "cusift_comm_create": (_i, [C.POINTER(_vp), _vp, _vp, _i, _i]),
    "cusift_comm_destroy": (_i, [_vp]),
    "cusift_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "cusift_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "cusift_expand_gathered": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "cusift_comm_library": (C.c_char_p, []),
    "cusift_comm_set_self_p2p": (_i, [_vp, _i]),
    "cusift_comm_use_library": (_i, [C.c_char_p]),
    "cusift_comm_ctx": (_vp, [_vp]),
    "cusift_comm_reserve": (_i, [_vp, _i, _i, _sz]),
    "cusift_comm_set_fixed_size": (_i, [_vp, _i]),
    "cusift_comm_set_wire_format": (_i, [_vp, _i]),
    "cusift_comm_host_waits": (C.c_ulonglong, [_vp]),
    "cusift_comm_host_wait_ms": (C.c_double, [_vp]),
    "cusift_comm_hip_syncs": (C.c_ulonglong, [_vp]),
    "cusift_allgatherv_begin": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz]),
    "cusift_allgatherv_finish": (_i, [_vp, _vp, _vp]),
    "cusift_allgatherv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "cusift_compact_gathered": (_i, [_vp, _vp, _sz, _i, _vp, _vp, _sz]),
    "cusift_tiled_plan": (_i, [_i, _i, _i, _i, _i, _i, _i] + [C.POINTER(_i)] * 9),
    "cusift_tiled_create": (_i, [C.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _PP, _i]),
    "cusift_tiled_destroy": (_i, [_vp]),
    "cusift_tiled_info": (_i, [_vp] + [C.POINTER(_i)] * 4),
    "cusift_tiled_band": (_i, [_vp, _i, C.POINTER(_vp)] + [C.POINTER(_i)] * 7),
    "cusift_tiled_load": (_i, [_vp, _vp, _i]),
    "cusift_tiled_build_octave": (_i, [_vp, _i]),
    "cusift_tiled_exchange": (_i, [_vp, _i]),
    "cusift_tiled_exchange_virtual": (_i, [C.POINTER(_vp), _i, _i]),
    "cusift_tiled_process": (_i, [_vp, _vp, _vp]),
    "cusift_tiled_extract": (_i, [_vp, _vp, _i, _vp, _vp]),
    "cusift_tiled_check": (_i, [_vp, C.POINTER(C.c_uint)]),
    "cusift_exchange_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cusift_exchange_halos": (_i, [_vp, _vp, _i, _i, _i, _i, _i]),
    "cusift_ctx_reserve": (_i, [_vp, _i, _i, _i, _PP]),
    "cusift_ctx_reserve_bands": (_i, [_vp, _i, _i]),
    "cusift_event_create": (_i, [_vp, C.POINTER(_vp)]),
    "cusift_event_record": (_i, [_vp, _vp]),
    "cusift_event_wait": (_i, [_vp, _vp]),
    "cusift_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "cusift_event_destroy": (_i, [_vp]),
    "cusift_pipe_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _PP, _i, _i, _sz]),
    "cusift_pipe_submit": (_i, [_vp, _vp, _i]),
    "cusift_pipe_collect": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int), C.POINTER(_sz)]),
    "cusift_pipe_in_flight": (_i, [_vp]),
    "cusift_pipe_destroy": (_i, [_vp]),
    "cusift_ctx_set_policy": (_i, [_vp, _i, _i]),
    "cusift_ctx_get_policy": (_i, [_vp, _i, C.POINTER(C.c_int)]),
    "cusift_ctx_arena_bytes": (_sz, [_vp]),
    "cusift_ctx_forks": (C.c_ulong, [_vp]),
    "cusift_ctx_timing_enable": (_i, [_vp, _i]),
    "cusift_ctx_timing_read": (_i, [_vp, C.POINTER(_f), C.POINTER(_i)]),
    "cusift_ctx_timing_reset": (_i, [_vp]),
    "cusift_kernel_occupancy": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "cusift_malloc": (_i, [C.POINTER(_vp), _sz]),
    "cusift_free": (_i, [_vp]),
    "cusift_memset": (_i, [_vp, _vp, _i, _sz]),
    "cusift_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "cusift_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "cusift_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "cusift_image_h2d": (_i, [_vp, _vp, _i, _vp, _i, _i]),
    "cusift_image_d2h": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "cusift_malloc_host": (_i, [C.POINTER(_vp), _sz]),
    "cusift_free_host": (_i, [_vp]),
    "cusift_image_u8_h2d": (_i, [_vp, _vp, _i, _vp, _i, _i]),
    "cusift_u8_to_f32": (_i, [_vp, _vp, _i, _sz, _vp, _i, _i, _i, _sz, _i]),
    "cusift_gaussian3x3": (_i, [_vp, _vp, _i, _sz, _vp, _i, _i, _i, _sz, _i, _f]),
    "cusift_scale_down": (_i, [_vp, _vp, _i, _sz, _vp, _i, _i, _i, _sz, _i, _f]),
    "cusift_scale_down_levels": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _vp, _vp, _i, _i, _f]),
    "cusift_extract_bands": (_i, [_vp, _vp, _i, _f, _f, _vp, _i, _vp, _i, _i, _vp]),
    "cusift_laplace_multi": (_i, [_vp, _vp, _i, _i, _i, _sz, _f, _vp, _sz, _i]),
    "cusift_laplace_taps": (_i, [_f, _vp]),
    "cusift_find_points_multi": (_i, [_vp, _vp, _i, _i, _i, _sz, _f, _f, _f, _vp, _i, _vp, _i]),
    "cusift_detect_multi": (_i, [_vp, _vp, _i, _i, _i, _sz, _f, _f, _f, _f, _vp, _i, _vp, _i]),
    "cusift_detect_multi_down": (_i, [_vp, _vp, _i, _i, _i, _sz, _f, _f, _f, _f, _vp, _i, _vp, _i, _vp, _i, _sz, _f]),
    "cusift_compute_orientations": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _vp, _vp, _i, _i]),
    "cusift_extract_descriptors": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _vp, _vp, _f, _i, _i]),
    "cusift_rootsift": (_i, [_vp, _vp, _i]),
    "cusift_math_eval": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz]),
    "cusift_scale_down_band": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _f]),
    "cusift_detect_band": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _i, _vp]),
    "cusift_describe_band": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f, _i, _i, _vp]),
    "cusift_match": (_i, [_vp, _vp, _i, _vp, _i, _i]),
    "cusift_memcpy2d_d2h": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "cusift_find_homography": (_i, [_vp, _vp, _i, _vp, _i, _f, _vp, C.POINTER(_i), _vp, _vp]),
    "cusift_pack_points": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "cusift_pack_points_compact": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "cusift_pack_points_trimmed": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "cusift_expand_trimmed": (_i, [_vp, _vp, _sz, _vp]),
    "cusift_expand_trimmed_host": (_i, [_vp, _sz, _vp]),
    "cusift_expand_points_host": (_i, [_vp, _sz, _vp]),
    "cusift_sort_points_host": (_i, [_vp, _i]),
    "cusift_extract_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _sz, _PP, _vp, _vp]),
    "cusift_graph_create": (_i, [_vp, C.POINTER(_vp), _vp, _i, _i, _i, _i, _sz, _PP, _vp, _vp]),
    "cusift_graph_launch": (_i, [_vp]),
    "cusift_graph_nodes": (_i, [_vp]),
    "cusift_graph_destroy": (_i, [_vp]),
    "cusift_extract": (_i, [_vp, _vp, _i, _i, _i, _PP, _vp, _vp, C.POINTER(_i)]),
    "cusift_extract_host": (_i, [_vp, _vp, _i, _i, _PP, _vp, _vp, C.POINTER(_i)]),
}

_lib = None


def _prefer_torch_hip_runtime():
    """A process can hold ONE libamdhip64.so.7.  libcusift_amd.so asks for it by soname and would pull in the system
    ROCm's; a PyTorch-ROCm wheel ships its own copy (plus matching HSA / comgr libraries) and stops seeing the GPU if
    it is imported after a different one was loaded (torch.cuda.is_available() == False).  So when torch is installed
    but not imported yet, its copy is loaded first -- the combination every torch-first program (bench.py,
    cusift_amd.batch) runs with anyway.  Without torch the system runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


def lib():
    """The loaded libcusift_amd.so with typed entry points. Raises if the extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CusiftError(
                "HIP extension %s is missing: run `python -m cusift_amd.build` (needs hipcc); "
                "there is no CPU fallback for the extraction path" % LIB_PATH
            )
        _prefer_torch_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != CUSIFT_OK:
        msg = lib().cusift_last_error()
        raise CusiftError("cusift error %d: %s" % (rc, msg.decode() if msg else "?"))


def default_params(**overrides):
    p = Params()
    lib().cusift_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown parameter %r" % k)
        setattr(p, k, v)
    return p


def device_count():
    n = C.c_int(0)
    check(lib().cusift_device_count(C.byref(n)))
    return n.value


def ialign_up(a, b):
    """iAlignUp, cutils.h:17"""
    return (a - a % b + b) if (a % b != 0) else a


# cusift_ctx_set_policy keys (include/cusift_amd.h)
POLICY_SIDE_STREAM, POLICY_OCTAVE_LISTS, POLICY_GENERIC_KERNELS, POLICY_LAUNCH_PER_OCTAVE, POLICY_MATCH_SPLITS, \
    POLICY_TILED_PER_OCTAVE, POLICY_PYRAMID_IN_DETECT = range(7)


class Context:
    """cusift_ctx: one device + one HIP stream + the scratch arena."""

    def __init__(self, device=0, stream=None):
        """stream=None: the context creates its own stream.  stream=<int handle>: borrow that hipStream_t;
        0 is the device's null stream (torch.cuda.current_stream().cuda_stream is 0 for the default stream)."""
        self._h = C.c_void_p()
        if stream is None:
            check(lib().cusift_ctx_create(C.byref(self._h), device, None))
        else:
            check(lib().cusift_ctx_create_borrowed(C.byref(self._h), device, C.c_void_p(int(stream))))
        self.device = device

    @property
    def handle(self):
        if not self._h:
            raise CusiftError("context already destroyed")
        return self._h

    def close(self):
        if self._h:
            lib().cusift_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        check(lib().cusift_ctx_synchronize(self.handle))

    def stream_handle(self):
        """The hipStream_t of the context as an integer (0 = the null stream): torch.cuda.ExternalStream(...) of it puts
        torch-side work on the stream the kernels run on."""
        return int(lib().cusift_ctx_stream(self.handle) or 0)

    def wait(self, other):
        """cusift_ctx_wait: work enqueued on this context from now on waits for what `other` has enqueued so far."""
        check(lib().cusift_ctx_wait(self.handle, other.handle))

    def reserve(self, n_images, w, h, params):
        check(lib().cusift_ctx_reserve(self.handle, n_images, w, h, C.byref(params)))

    def arena_bytes(self):
        return lib().cusift_ctx_arena_bytes(self.handle)

    def set_policy(self, key, value):
        check(lib().cusift_ctx_set_policy(self.handle, int(key), int(value)))

    def get_policy(self, key):
        v = C.c_int(0)
        check(lib().cusift_ctx_get_policy(self.handle, int(key), C.byref(v)))
        return v.value

    def forks(self):
        """cusift_ctx_forks: extractions that ran octave 0's detection on the context's second stream."""
        return int(lib().cusift_ctx_forks(self.handle))

    # ---- timing ----
    def timing_enable(self, on=True):
        check(lib().cusift_ctx_timing_enable(self.handle, 1 if on else 0))

    def timing_reset(self):
        check(lib().cusift_ctx_timing_reset(self.handle))

    def timing_read(self):
        ms = (C.c_float * NUM_STAGES)()
        n = (C.c_int * NUM_STAGES)()
        check(lib().cusift_ctx_timing_read(self.handle, ms, n))
        return {STAGE_NAMES[i]: (float(ms[i]), int(n[i])) for i in range(NUM_STAGES)}

    # ---- raw device memory ----
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib().cusift_malloc(C.byref(p), nbytes))
        return p.value

    def free(self, ptr):
        check(lib().cusift_free(C.c_void_p(ptr)))

    def memset(self, ptr, value, nbytes):
        check(lib().cusift_memset(self.handle, C.c_void_p(ptr), value, nbytes))

    def h2d(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        check(lib().cusift_memcpy_h2d(self.handle, C.c_void_p(d_ptr), arr.ctypes.data, arr.nbytes))

    def d2h(self, arr, d_ptr):
        assert arr.flags["C_CONTIGUOUS"]
        check(lib().cusift_memcpy_d2h(self.handle, arr.ctypes.data, C.c_void_p(d_ptr), arr.nbytes))

    # ---- front-end ----
    def image_u8_h2d(self, d_dst, dst_pitch, img_u8):
        img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
        h, w = img_u8.shape
        check(lib().cusift_image_u8_h2d(self.handle, d_dst, dst_pitch, img_u8.ctypes.data, w, h))

    def u8_to_f32(self, d_dst, dst_pitch, d_src, w, h, src_pitch_bytes, n_images=1, dst_stride=None, src_stride=None):
        dst_stride = h * dst_pitch if dst_stride is None else dst_stride
        src_stride = h * src_pitch_bytes if src_stride is None else src_stride
        check(lib().cusift_u8_to_f32(self.handle, d_dst, dst_pitch, dst_stride, d_src, w, h, src_pitch_bytes,
                                     src_stride, n_images))

    def gaussian3x3(self, d_dst, dst_pitch, d_src, w, h, src_pitch, sigma, n_images=1, dst_stride=None,
                    src_stride=None):
        dst_stride = h * dst_pitch if dst_stride is None else dst_stride
        src_stride = h * src_pitch if src_stride is None else src_stride
        check(lib().cusift_gaussian3x3(self.handle, d_dst, dst_pitch, dst_stride, d_src, w, h, src_pitch, src_stride,
                                       n_images, sigma))

    # ---- stage entry points (device pointers are plain ints) ----
    def scale_down(self, d_dst, dst_pitch, d_src, w, h, src_pitch, n_images=1, dst_stride=None, src_stride=None,
                   variance=0.5):
        dst_stride = (h // 2) * dst_pitch if dst_stride is None else dst_stride
        src_stride = h * src_pitch if src_stride is None else src_stride
        check(lib().cusift_scale_down(self.handle, d_dst, dst_pitch, dst_stride, d_src, w, h, src_pitch, src_stride,
                                      n_images, variance))

    def scale_down_levels(self, d_src, w, h, src_pitch, d_levels, pitches, n_images=1, src_stride=None, strides=None,
                          variance=0.5):
        """cusift_scale_down_levels: the ScaleDown chain (level k = (w >> k) x (h >> k), k = 1..len(d_levels)) in one launch."""
        n = len(d_levels)
        src_stride = h * src_pitch if src_stride is None else src_stride
        if strides is None:
            strides = [(h >> (k + 1)) * pitches[k] for k in range(n)]
        ptrs = (C.c_void_p * n)(*[int(p) for p in d_levels])
        pit = (C.c_int * n)(*[int(p) for p in pitches])
        strd = (C.c_size_t * n)(*[int(x) for x in strides])
        check(lib().cusift_scale_down_levels(self.handle, d_src, w, h, src_pitch, src_stride, ptrs, pit, strd, n,
                                             n_images, variance))

    def laplace_multi(self, d_img, w, h, pitch, init_blur, d_dog, n_images=1, img_stride=None, dog_stride=None):
        img_stride = h * pitch if img_stride is None else img_stride
        dog_stride = 7 * h * pitch if dog_stride is None else dog_stride
        check(lib().cusift_laplace_multi(self.handle, d_img, w, h, pitch, img_stride, init_blur, d_dog, dog_stride,
                                         n_images))

    def find_points_multi(self, d_dog, w, h, pitch, peak_thresh, edge_thresh, subsampling, d_points, max_pts,
                          d_counters, n_images=1, dog_stride=None):
        dog_stride = 7 * h * pitch if dog_stride is None else dog_stride
        check(lib().cusift_find_points_multi(self.handle, d_dog, w, h, pitch, dog_stride, peak_thresh, edge_thresh,
                                             subsampling, d_points, max_pts, d_counters, n_images))

    def detect_multi(self, d_img, w, h, pitch, init_blur, peak_thresh, edge_thresh, subsampling, d_points, max_pts,
                     d_counters, n_images=1, img_stride=None):
        img_stride = h * pitch if img_stride is None else img_stride
        check(lib().cusift_detect_multi(self.handle, d_img, w, h, pitch, img_stride, init_blur, peak_thresh,
                                        edge_thresh, subsampling, d_points, max_pts, d_counters, n_images))

    def detect_multi_down(self, d_img, w, h, pitch, init_blur, peak_thresh, edge_thresh, subsampling, d_heads, max_pts,
                          d_counters, d_next, next_pitch, n_images=1, img_stride=None, next_stride=None, variance=0.5):
        """cusift_detect_multi_down: the fused detection (keypoint HEADS, 64 bytes each) + the next octave's image."""
        img_stride = h * pitch if img_stride is None else img_stride
        next_stride = (h // 2) * next_pitch if next_stride is None else next_stride
        check(lib().cusift_detect_multi_down(self.handle, d_img, w, h, pitch, img_stride, init_blur, peak_thresh,
                                             edge_thresh, subsampling, d_heads, max_pts, d_counters, n_images, d_next,
                                             next_pitch, next_stride, variance))

    def compute_orientations(self, d_img, w, h, pitch, d_points, max_pts, d_first, d_counters, tex_frac_bits=8,
                             n_images=1, img_stride=None):
        img_stride = h * pitch if img_stride is None else img_stride
        check(lib().cusift_compute_orientations(self.handle, d_img, w, h, pitch, img_stride, d_points, max_pts,
                                                d_first, d_counters, tex_frac_bits, n_images))

    def extract_descriptors(self, d_img, w, h, pitch, d_points, max_pts, d_first, d_counters, subsampling,
                            tex_frac_bits=8, n_images=1, img_stride=None):
        img_stride = h * pitch if img_stride is None else img_stride
        check(lib().cusift_extract_descriptors(self.handle, d_img, w, h, pitch, img_stride, d_points, max_pts,
                                               d_first, d_counters, subsampling, tex_frac_bits, n_images))

    # ---- band forms (strip tiling) ----
    def scale_down_band(self, d_dst, dst_pitch, dst_row0, r_begin, r_end, d_src, w, h_src, src_pitch, src_row0,
                        h_src_global, variance=0.5):
        check(lib().cusift_scale_down_band(self.handle, d_dst, dst_pitch, dst_row0, r_begin, r_end, d_src, w, h_src,
                                           src_pitch, src_row0, h_src_global, variance))

    def detect_band(self, d_img, w, h, pitch, row0, h_global, cy_begin, cy_end, init_blur, peak_thresh, edge_thresh,
                    subsampling, d_points, max_pts, d_counter):
        check(lib().cusift_detect_band(self.handle, d_img, w, h, pitch, row0, h_global, cy_begin, cy_end, init_blur,
                                       peak_thresh, edge_thresh, subsampling, d_points, max_pts, d_counter))

    def describe_band(self, d_img, w, h, pitch, row0, h_global, d_points, max_pts, d_first, d_counter, subsampling,
                      tex_frac_bits=8, d_flags=None, root_sift=0):
        check(lib().cusift_describe_band(self.handle, d_img, w, h, pitch, row0, h_global, d_points, max_pts, d_first,
                                         d_counter, subsampling, tex_frac_bits, root_sift, d_flags))

    def math_eval(self, op, d_a, d_b, d_out, d_out2, n):
        """cusift_math_eval: op 0 expf, 1 exp2f, 2 atan2f(a, b), 3 sincosf -> (out, out2), 4 the descriptor's angle
        coordinate of (dy = a, dx = b), 5 the orientation bin's shortcut of (dy = a, dx = b): out = bin (+ 64 if the sample
        is `near` a bin edge: the kernel then evaluates the reference's formula), out2 = that formula's bin; device
        pointers."""
        check(lib().cusift_math_eval(self.handle, op, d_a, d_b, d_out, d_out2, n))

    def rootsift(self, d_points, num_pts):
        check(lib().cusift_rootsift(self.handle, d_points, num_pts))

    def pack_points(self, d_points, d_counters, n_images, max_pts, d_packed, capacity, d_offsets=None):
        check(lib().cusift_pack_points(self.handle, d_points, d_counters, n_images, max_pts, d_packed, capacity,
                                       d_offsets))

    # ---- matcher ----
    def pack_points_trimmed(self, d_points, d_counters, n_images, max_pts, d_packed, capacity, d_offsets=None):
        check(lib().cusift_pack_points_trimmed(self.handle, d_points, d_counters, n_images, max_pts, d_packed, capacity,
                                               d_offsets))

    def expand_trimmed(self, d_trimmed, n, d_points):
        check(lib().cusift_expand_trimmed(self.handle, d_trimmed, n, d_points))

    def pack_points_compact(self, d_points, d_counters, n_images, max_pts, d_packed, capacity, d_offsets=None):
        check(lib().cusift_pack_points_compact(self.handle, d_points, d_counters, n_images, max_pts, d_packed, capacity,
                                               d_offsets))

    def match(self, d_sift1, n1, d_sift2, n2, distance=1):
        """MatchSiftData on device records: distance 1 = L2 (2 - 2 x.y), 0 = dot product."""
        check(lib().cusift_match(self.handle, d_sift1, n1, d_sift2, n2, distance))

    def find_homography(self, d_sift, num_pts, rand_pts, thresh=5.0, want_all=False):
        """cusift_find_homography: rand_pts is an int32 array [4, num_loops] of sample indices into d_sift.
        Returns (H[9], num_matches) or, with want_all, (H, num_matches, all_homographies[8, L], all_counts[L])."""
        rand_pts = np.ascontiguousarray(rand_pts, dtype=np.int32)
        assert rand_pts.ndim == 2 and rand_pts.shape[0] == 4, rand_pts.shape
        loops = rand_pts.shape[1]
        hom = np.zeros(9, dtype=np.float32)
        n = C.c_int(0)
        all_h = np.zeros((8, loops), dtype=np.float32) if want_all else None
        all_c = np.zeros(loops, dtype=np.int32) if want_all else None
        check(lib().cusift_find_homography(self.handle, d_sift, num_pts, rand_pts.ctypes.data, loops, thresh,
                                           hom.ctypes.data, C.byref(n), all_h.ctypes.data if want_all else None,
                                           all_c.ctypes.data if want_all else None))
        return (hom, n.value, all_h, all_c) if want_all else (hom, n.value)

    # ---- drivers ----
    def extract_batch(self, d_imgs, n_images, w, h, pitch, image_stride, params, d_points, d_counters):
        check(lib().cusift_extract_batch(self.handle, d_imgs, n_images, w, h, pitch, image_stride, C.byref(params),
                                         d_points, d_counters))

    def record_graph(self, d_imgs, n_images, w, h, pitch, image_stride, params, d_points, d_counters):
        """cusift_graph_create: extract_batch for fixed buffers, recorded once as a hipGraph (see ExtractGraph)."""
        return ExtractGraph(self, d_imgs, n_images, w, h, pitch, image_stride, params, d_points, d_counters)

    def extract(self, d_img, w, h, pitch, params, d_points, h_points=None):
        n = C.c_int(0)
        hp = h_points.ctypes.data if h_points is not None else None
        check(lib().cusift_extract(self.handle, d_img, w, h, pitch, C.byref(params), d_points, hp, C.byref(n)))
        return n.value

    def extract_host(self, img, params, d_points, h_points=None):
        img = np.ascontiguousarray(img, dtype=np.float32)
        h, w = img.shape
        n = C.c_int(0)
        hp = h_points.ctypes.data if h_points is not None else None
        check(lib().cusift_extract_host(self.handle, img.ctypes.data, w, h, C.byref(params), d_points, hp,
                                        C.byref(n)))
        return n.value


UNIQUE_ID_BYTES = 128


def comm_use_library(path=None):
    """cusift_comm_use_library: the library (RCCL, or anything exporting the same nine nccl* entry points) that
    comm_unique_id() / Comm() bind from now on; None = the default search (next to the process's HIP runtime)."""
    check(lib().cusift_comm_use_library(path.encode() if path else None))


def comm_unique_id():
    """cusift_comm_get_unique_id: 128 opaque bytes; rank 0 makes them, every rank passes them to Comm()."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(lib().cusift_comm_get_unique_id(buf))
    return buf.raw


class Comm:
    """cusift_comm: an RCCL communicator bound to a Context (its device and stream); one process per GPU.
    The all-gatherv of SiftData and the row / halo exchange of the strip tiling (include/cusift_amd.h)."""

    def __init__(self, ctx, unique_id, rank, world, self_p2p=False):
        assert len(unique_id) == UNIQUE_ID_BYTES
        self._h = C.c_void_p()
        self.ctx = ctx  # keeps the context alive
        self.rank, self.world = rank, world
        self._slots = []  # n_images_max of the exchanges in flight, oldest first
        check(lib().cusift_comm_create(C.byref(self._h), ctx.handle, unique_id, rank, world))
        if self_p2p:
            check(lib().cusift_comm_set_self_p2p(self._h, 1))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().cusift_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def library():
        p = lib().cusift_comm_library()
        return p.decode() if p else ""

    def info(self):
        """cusift_comm_info: what the bound LIBRARY says about this communicator -- {"lib_ranks": ncclCommCount,
        "lib_rank": ncclCommUserRank, "lib_version": ncclGetVersion}; -1 where the library lacks the call."""
        n, r, v = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        check(lib().cusift_comm_info(self._h, C.byref(n), C.byref(r), C.byref(v)))
        return {"lib_ranks": n.value, "lib_rank": r.value, "lib_version": v.value}

    def expand_gathered(self, d_gathered_trimmed, region_cap, totals, d_points):
        """cusift_expand_gathered: regions of 540-byte trimmed records -> regions of 588-byte SiftPoint records, one
        launch on the communicator's stream behind the exchange (asynchronous)."""
        t = np.ascontiguousarray(np.asarray(totals, dtype=np.uint64))
        assert len(t) == self.world
        check(lib().cusift_expand_gathered(self._h, d_gathered_trimmed, region_cap, t.ctypes.data, d_points))

    def reserve(self, n_images_max, tickets=4, stage_records=0):
        check(lib().cusift_comm_reserve(self._h, n_images_max, tickets, stage_records))

    def set_fixed_size(self, on=True):
        check(lib().cusift_comm_set_fixed_size(self._h, 1 if on else 0))

    def set_wire_format(self, fmt=True):
        """fmt: "exact" / "compact" / "trimmed" (or the number; True / False = compact / exact)."""
        if isinstance(fmt, str):
            fmt = WIRE_FORMATS[fmt][0]
        check(lib().cusift_comm_set_wire_format(self._h, int(fmt)))

    def host_waits(self):
        return int(lib().cusift_comm_host_waits(self._h))

    def host_wait_ms(self):
        return float(lib().cusift_comm_host_wait_ms(self._h))

    def hip_syncs(self):
        return int(lib().cusift_comm_hip_syncs(self._h))

    def allgatherv_begin(self, d_points, d_counters, n_images, max_pts, n_images_max, d_gathered, region_cap,
                         producer=None):
        """`producer`: the Context that extracted d_points (the exchange is ordered after it; None: the caller has)."""
        check(lib().cusift_allgatherv_begin(self._h, producer.handle if producer is not None else None, d_points,
                                            d_counters, n_images, max_pts, n_images_max, d_gathered, region_cap))
        self._slots.append(n_images_max)

    def allgatherv_finish(self):
        """Completes the OLDEST begin.  Returns (counts uint32 [world, n_images_max], totals uint64 [world]) -- host
        arrays; rank r's records are d_gathered[r * region_cap : r * region_cap + totals[r]]."""
        if not self._slots:
            raise CusiftError("allgatherv: finish() without begin()")
        counts = np.zeros((self.world, self._slots[0]), dtype=np.uint32)
        totals = np.zeros(self.world, dtype=np.uint64)
        try:
            check(lib().cusift_allgatherv_finish(self._h, counts.ctypes.data, totals.ctypes.data))
        finally:
            self._slots.pop(0)
        return counts, totals

    def allgatherv(self, d_points, d_counters, n_images, max_pts, n_images_max, d_gathered, region_cap, producer=None):
        self.allgatherv_begin(d_points, d_counters, n_images, max_pts, n_images_max, d_gathered, region_cap, producer)
        return self.allgatherv_finish()

    def exchange_rows(self, d_band, pitch, band_rows, ops):
        """ops: list of (peer, send_row, send_rows, recv_row, recv_rows); d_band holds band_rows rows."""
        a = np.ascontiguousarray(np.array(ops, dtype=np.int32).reshape(-1, 5).T)
        n = a.shape[1]
        check(lib().cusift_exchange_rows(self._h, d_band, pitch, band_rows, n, a[0].ctypes.data, a[1].ctypes.data,
                                         a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data))

    def exchange_halos(self, d_band, pitch, top_halo, own_rows, bottom_halo, send_rows):
        check(lib().cusift_exchange_halos(self._h, d_band, pitch, top_halo, own_rows, bottom_halo, send_rows))


def compact_gathered(ctx, d_gathered, region_cap, totals, d_out, capacity):
    """cusift_compact_gathered: the regions of an all-gatherv back to back in rank order (asynchronous on ctx)."""
    t = np.ascontiguousarray(totals, dtype=np.uint64)
    check(lib().cusift_compact_gathered(ctx.handle, d_gathered, region_cap, len(t), t.ctypes.data, d_out, capacity))


def tiled_plan(W, H, world, num_octaves, halo=0, rank=0, octave=0):
    """cusift_tiled_plan (host only): dict of the strip plan's geometry for (rank, octave)."""
    v = [C.c_int(0) for _ in range(9)]
    check(lib().cusift_tiled_plan(W, H, world, num_octaves, halo, rank, octave, *[C.byref(x) for x in v]))
    keys = ("n_octaves", "collapse", "w", "h", "pitch", "own_begin", "own_end", "band_begin", "band_end")
    return dict(zip(keys, (x.value for x in v)))


class Tiled:
    """cusift_tiled: one rank of the strip-tiled extraction of ONE large image (BASELINE configs[4])."""

    def __init__(self, ctx, comm, rank, world, W, H, params, halo=0):
        self._h = C.c_void_p()
        self.ctx, self.comm = ctx, comm  # kept alive
        self.rank, self.world = rank, world
        check(lib().cusift_tiled_create(C.byref(self._h), ctx.handle, comm.handle if comm is not None else None, rank,
                                        world, W, H, C.byref(params), halo))
        v = [C.c_int(0) for _ in range(4)]
        check(lib().cusift_tiled_info(self._h, *[C.byref(x) for x in v]))
        self.n_oct, self.collapse, self.root, self.halo = (x.value for x in v)

    @property
    def handle(self):
        return self._h

    def band(self, octave):
        """(device pointer or None, dict of geometry) of this rank's band of `octave`."""
        ptr = C.c_void_p()
        v = [C.c_int(0) for _ in range(7)]
        check(lib().cusift_tiled_band(self._h, octave, C.byref(ptr), *[C.byref(x) for x in v]))
        keys = ("w", "h", "pitch", "own_begin", "own_end", "band_begin", "band_end")
        return ptr.value, dict(zip(keys, (x.value for x in v)))

    def load(self, d_strip, strip_pitch):
        check(lib().cusift_tiled_load(self._h, d_strip, strip_pitch))

    def build_octave(self, o):
        check(lib().cusift_tiled_build_octave(self._h, o))

    def exchange(self, o):
        check(lib().cusift_tiled_exchange(self._h, o))

    def process(self, d_points, d_counter):
        check(lib().cusift_tiled_process(self._h, d_points, d_counter))

    def extract(self, d_strip, strip_pitch, d_points, d_counter):
        check(lib().cusift_tiled_extract(self._h, d_strip, strip_pitch, d_points, d_counter))

    def check(self, strict=True):
        """Blocking.  Number of keypoints whose footprint left the halo; raises if non-zero and strict."""
        f = C.c_uint(0)
        rc = lib().cusift_tiled_check(self._h, C.byref(f))
        if rc != CUSIFT_OK and (strict or f.value == 0):
            check(rc)
        return f.value

    def close(self):
        if self._h:
            lib().cusift_tiled_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tiled_exchange_virtual(tiles, octave):
    """cusift_tiled_exchange_virtual: the exchange of `octave` among all ranks' Tiled objects of ONE process."""
    arr = (C.c_void_p * len(tiles))(*[t.handle for t in tiles])
    check(lib().cusift_tiled_exchange_virtual(arr, len(tiles), octave))


class ExtractGraph:
    """A recorded extract_batch (cusift_graph_*): launch() replays all of its kernels with one host call."""

    def __init__(self, ctx, d_imgs, n_images, w, h, pitch, image_stride, params, d_points, d_counters):
        self._g = C.c_void_p()
        self.ctx = ctx  # keeps the context alive
        check(lib().cusift_graph_create(ctx.handle, C.byref(self._g), d_imgs, n_images, w, h, pitch, image_stride,
                                        C.byref(params), d_points, d_counters))

    @property
    def nodes(self):
        return lib().cusift_graph_nodes(self._g)

    def launch(self):
        check(lib().cusift_graph_launch(self._g))

    def close(self):
        if self._g:
            lib().cusift_graph_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kernel_occupancy(name):
    """(resident workgroups per CU, threads per workgroup) of a named kernel; needs a GPU."""
    n, t = C.c_int(0), C.c_int(0)
    check(lib().cusift_kernel_occupancy(name.encode(), C.byref(n), C.byref(t)))
    return n.value, t.value


def laplace_taps(init_blur):
    taps = np.zeros(8 * 16, dtype=np.float32)
    check(lib().cusift_laplace_taps(init_blur, taps.ctypes.data))
    return taps


class DeviceBuffer:
    """A raw HBM allocation made through the C ABI (cusift_malloc / cusift_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr = ctx.malloc(self.nbytes)

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, arr.nbytes)
        ctx.h2d(buf.ptr, arr)
        return buf

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        self.ctx.d2h(out, self.ptr)
        return out

    def zero(self):
        self.ctx.memset(self.ptr, 0, self.nbytes)

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Pinned host memory made through the C ABI (cusift_malloc_host / cusift_free_host): what include/cuSIFT.h's
    SiftData keeps its host records in -- a read-back into it is one DMA, not a staged copy."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().cusift_malloc_host(C.byref(p), self.nbytes))
        self.ptr = p.value

    def as_numpy(self, dtype, count):
        dtype = np.dtype(dtype)
        assert count * dtype.itemsize <= self.nbytes
        raw = (C.c_char * (count * dtype.itemsize)).from_address(self.ptr)
        return np.frombuffer(raw, dtype=dtype, count=count)  # a view: valid until free()

    def free(self):
        if self.ptr:
            check(lib().cusift_free_host(C.c_void_p(self.ptr)))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


PIPE_U8, PIPE_F32 = 0, 1


class Pipe:
    """cusift_pipe: host frames in, SiftData in pinned host memory out, `depth` batches in flight (C ABI; no torch)."""

    def __init__(self, device, n_images, w, h, params, input_format=PIPE_U8, depth=3, records_capacity=0):
        self._h = C.c_void_p()
        self.n_images, self.w, self.h, self.format = n_images, w, h, input_format
        check(lib().cusift_pipe_create(C.byref(self._h), device, n_images, w, h, C.byref(params), input_format, depth,
                                       records_capacity))

    def submit(self, frames):
        """frames: C-contiguous [n, h, w] uint8 / float32 array (or anything with ctypes.data and the same layout,
        e.g. a pinned torch tensor's numpy view).  It must stay alive and untouched until collected."""
        a = frames
        want = np.uint8 if self.format == PIPE_U8 else np.float32
        assert a.dtype == want and a.ndim == 3 and a.shape[1:] == (self.h, self.w) and a.flags["C_CONTIGUOUS"], a.shape
        check(lib().cusift_pipe_submit(self._h, a.ctypes.data, int(a.shape[0])))

    def collect(self):
        """-> (records: SIFT_POINT_DTYPE view of the pinned slot, offsets [n + 1]); valid until the next collect()."""
        rec, off = C.c_void_p(), C.c_void_p()
        n, total = C.c_int(0), C.c_size_t(0)
        check(lib().cusift_pipe_collect(self._h, C.byref(rec), C.byref(off), C.byref(n), C.byref(total)))
        offsets = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint32)), shape=(n.value + 1,))
        if total.value == 0:
            return np.zeros(0, dtype=SIFT_POINT_DTYPE), offsets
        raw = np.ctypeslib.as_array(C.cast(rec, C.POINTER(C.c_uint8)), shape=(total.value * SIFT_POINT_BYTES,))
        return raw.view(SIFT_POINT_DTYPE), offsets

    def in_flight(self):
        return int(lib().cusift_pipe_in_flight(self._h))

    def close(self):
        if self._h:
            lib().cusift_pipe_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def expand_trimmed(trimmed):
    """cusift_expand_trimmed_host: 540-byte trimmed records -> SiftPoint records (exact; unwritten fields zero)."""
    trimmed = np.ascontiguousarray(trimmed)
    assert trimmed.dtype == TRIMMED_POINT_DTYPE
    out = np.zeros(len(trimmed), dtype=SIFT_POINT_DTYPE)
    check(lib().cusift_expand_trimmed_host(trimmed.ctypes.data, len(trimmed), out.ctypes.data))
    return out


def expand_points(compact):
    """cusift_expand_points_host: COMPACT_POINT_DTYPE array -> SIFT_POINT_DTYPE array (host)."""
    compact = np.ascontiguousarray(compact)
    assert compact.dtype == COMPACT_POINT_DTYPE
    out = np.zeros(len(compact), dtype=SIFT_POINT_DTYPE)
    check(lib().cusift_expand_points_host(compact.ctypes.data, len(compact), out.ctypes.data))
    return out


def sort_points(points):
    """cusift_sort_points_host: canonical order (octave coarsest first, then y, x, scale) of a host SiftPoint array,
    in place; equal point sets give equal arrays."""
    assert points.dtype == SIFT_POINT_DTYPE and points.flags["C_CONTIGUOUS"]
    check(lib().cusift_sort_points_host(points.ctypes.data, len(points)))
    return points


def match_filter(points, score_threshold=999.0, ambiguity_threshold=1.0):
    """The host-side filter of MatchSiftData (extras/matching.cu:318-349, MatchType2D): indices of the points
    whose score < scoreThreshold^2 and ambiguity < ambiguityThreshold^2."""
    t2 = np.float32(score_threshold) * np.float32(score_threshold)
    a2 = np.float32(ambiguity_threshold) * np.float32(ambiguity_threshold)
    return np.nonzero((points["score"] < t2) & (points["ambiguity"] < a2))[0]
