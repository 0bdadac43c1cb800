"""cusift_amd -- MI355X-native SIFT extraction (the cuSIFT hot path) behind a C ABI.

Layout
    csrc/       hand-written gfx950 HIP kernels + the C ABI implementation (libcusift_amd.so)
    capi.py     ctypes binding of include/cusift_amd.h (Context, DeviceBuffer, ExtractGraph, ...)
    batch.py    batched extraction on torch device tensors (HBM-resident inputs, one stream)
    dist.py     one-process-per-GPU sharding of an image batch + all-gatherv of SiftData (RCCL)
    tiling.py   one large image strip-tiled over several GPUs with per-octave halo exchange
    synth.py    seeded synthetic inputs of the benchmark configurations
    build.py    hipcc build of the shared object (in-tree)
The host-side mirror of the reference's C++ surface (SiftData / cuImage / ExtractSift, MatchSiftData, FindHomography
...) is C++ like the reference: include/cuSIFT.h, matching.h, debug.h, homography.h over the same C ABI.
"""
from .capi import (SIFT_POINT_BYTES, SIFT_POINT_DTYPE, Context, CusiftError, DeviceBuffer, Params,  # noqa: F401
                   default_params, device_count, ialign_up)

__version__ = "0.1.0"
