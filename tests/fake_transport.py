"""Locates (and, if need be, builds) tests/fake_rccl/libfake_rccl.so -- the test-only in-process stand-in for librccl
(see tests/fake_rccl/fake_rccl.cpp) -- and reads its call log."""
import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl", "fake_rccl.cpp")
LIB = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = "%s.%d.tmp" % (LIB, os.getpid())
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", SRC, "-o", tmp])
    os.replace(tmp, LIB)
    return LIB


def fake_rccl_path():
    return build()


def fake_stats():
    """The transport's call log since it was loaded (zeros before)."""
    out = (C.c_ulonglong * 8)()
    C.CDLL(fake_rccl_path()).fake_rccl_stats(out)
    keys = ("groups", "sends", "recvs", "allgathers", "bytes", "mismatches", "timeouts", "comms")
    return dict(zip(keys, (int(x) for x in out)))
