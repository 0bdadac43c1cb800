"""The Python-free build: the top-level Makefile is the one recipe (python -m cusift_amd.build and CMakeLists.txt both
drive it); an installed tree is found by find_package(cusift_amd) and a consumer links the reference's detector test
against it with its ordinary C++ compiler (the reference is a CMake project: CMakeLists.txt:45-72)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_build_is_the_makefile():
    from cusift_amd import build as b

    lib = b.build()
    assert lib == os.path.join(ROOT, "cusift_amd", "libcusift_amd.so") and os.path.exists(lib)
    assert not b.is_stale()  # `make -q all` agrees
    text = open(os.path.join(ROOT, "Makefile")).read()
    for flag in ("-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=$(ARCH)"):
        assert flag in text
    for flag in b.HIPCC_FLAGS:  # the tools' copy of the flags says what the Makefile says
        assert flag in ("-shared",) or flag.replace("gfx950", "$(ARCH)") in text, flag


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_install_and_find_package(tmp_path):
    prefix = tmp_path / "prefix"
    subprocess.check_call(["make", "-C", ROOT, "install", "PREFIX=%s" % prefix], stdout=subprocess.DEVNULL)
    assert (prefix / "lib" / "libcusift_amd.so").exists() and (prefix / "include" / "cusift_amd" / "cuSIFT.h").exists()
    assert (prefix / "lib" / "cmake" / "cusift_amd" / "cusift_amdConfig.cmake").exists()
    src = tmp_path / "consumer"
    src.mkdir()
    shutil.copy(os.path.join(ROOT, "tests", "cpp", "detector_dropin.cpp"), src / "detector.cpp")
    (src / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.16)\nproject(consumer LANGUAGES CXX)\n"
        "find_package(cusift_amd REQUIRED)\nadd_executable(detector detector.cpp)\n"
        "target_compile_features(detector PRIVATE cxx_std_14)\n"
        "target_link_libraries(detector PRIVATE cusift_amd::cusift_amd)\n")
    bld = tmp_path / "build"
    subprocess.check_call(["cmake", "-S", str(src), "-B", str(bld), "-DCMAKE_PREFIX_PATH=%s" % prefix],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(bld)], stdout=subprocess.DEVNULL)
    assert (bld / "detector").exists()
    out = subprocess.run(["ldd", str(bld / "detector")], capture_output=True, text=True).stdout
    assert "libcusift_amd.so" in out
