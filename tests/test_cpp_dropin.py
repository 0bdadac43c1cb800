"""The C++ drop-in surface (include/cuSIFT.h): a re-write of the reference's own detector test
(test/detector.cpp:18-90) plus the legacy ExtractSift trio, compiled with plain g++."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "detector_dropin")
BIN_MATCH = os.path.join(CPP, "matching_dropin")
BIN_HOMO = os.path.join(CPP, "homography_dropin")
BIN_MULTI = os.path.join(CPP, "multigpu_dropin")
BIN_PIPE = os.path.join(CPP, "pipeline_dropin")
BIN_TILED = os.path.join(CPP, "tiled_dropin")
BIN_HOSTPIPE = os.path.join(CPP, "hostpipe_dropin")
BIN_SCALING = os.path.join(CPP, "scaling_bench")
BIN_THREADS = os.path.join(CPP, "threads_dropin")


def build():
    subprocess.check_call(["make", "-C", CPP, "all"], stdout=subprocess.DEVNULL)
    assert os.path.exists(BIN) and os.path.exists(BIN_MATCH) and os.path.exists(BIN_HOMO) and os.path.exists(BIN_MULTI)
    assert os.path.exists(BIN_PIPE) and os.path.exists(BIN_TILED) and os.path.exists(BIN_HOSTPIPE)
    assert os.path.exists(BIN_THREADS)


def test_dropin_header_compiles_and_links_with_gxx():
    """No HIP/CUDA headers on the include path: cuSIFT.h + cusift_amd.h must be self-contained C++."""
    for b in (BIN, BIN_MATCH, BIN_HOMO, BIN_MULTI, BIN_PIPE, BIN_TILED, BIN_HOSTPIPE, BIN_SCALING, BIN_THREADS):
        if os.path.exists(b):
            os.remove(b)
    build()


def test_dropin_header_exports_the_reference_surface():
    text = open(os.path.join(ROOT, "include", "cuSIFT.h")).read()
    for needle in ("class SiftData", "class SiftPoint", "class cuImage", "ExtractSift(", "InitSiftData(",
                   "FreeSiftData(", "ScaleDown(", "InitCuda(", "ConvertSiftToRootSift", "Synchronize"):
        assert needle in text, needle
    assert "#include <hip" not in text and "cuda_runtime" not in text  # plain C++ over the C ABI
    assert "ExtractRootSift(" in text
    dbg = open(os.path.join(ROOT, "include", "debug.h")).read()
    for needle in ("AddSiftData(", "ReadVLFeatSiftData(", "WriteVLFeatSiftData(", "ReadMATLABMatchIndices(",
                   "PrintSiftData("):
        assert needle in dbg, needle
    assert "#include <hip" not in dbg and "opencv" not in dbg.replace("OpenCV", "")


@pytest.mark.gpu
def test_dropin_detector_program_passes_on_gpu():
    build()  # incremental: rebuilds when a header changed (cusift_params is passed by pointer -- ABI)
    out = subprocess.run([BIN, os.path.join(ROOT, "tests", "golden", "gray1.pgm"),
                          os.path.join(ROOT, "tests", "golden", "cusift1_check.bin")], capture_output=True, text=True,
                         timeout=300)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PASSED" in out.stdout
    assert "num pts: golden 4096, extracted 4096" in out.stdout
    assert "Total time incl memory" in out.stdout  # the reference prints this on every call (cuSIFT.cu:117-119)
    assert "coarse block 1555 matched, 1555 bit-identical descriptors" in out.stdout  # ExtractRootSift


@pytest.mark.gpu
def test_dropin_matching_program_passes_on_gpu():
    """extras/matching.h surface: the reference's MatchingTest + MatchingRatioTest (test/test.cpp:25-56)."""
    build()
    g = os.path.join(ROOT, "tests", "golden")
    out = subprocess.run([BIN_MATCH, os.path.join(g, "vlfeat_sift1.bin"), os.path.join(g, "vlfeat_sift2.bin"),
                          os.path.join(g, "match_indices1_2.bin")], capture_output=True, text=True, timeout=300)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "326 / 326 agree" in out.stdout and "ratio test: 340 matches" in out.stdout
    # extras/debug.h surface: AddSiftData doubling (884 -> 2652 points, 1024 -> 4096 slots), dump round trip
    assert "AddSiftData: 884 -> 2652 points, capacity 1024 -> 4096" in out.stdout
    assert "dump round trip: 2652 / 2652 records identical" in out.stdout


@pytest.mark.gpu
def test_dropin_homography_program_passes_on_gpu():
    """extras/homography.h surface: FindHomography (RANSAC on the GPU) + ImproveHomography on planted matches."""
    build()
    out = subprocess.run([BIN_HOMO], capture_output=True, text=True, timeout=300)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PASSED" in out.stdout and "FindHomography:" in out.stdout and "ImproveHomography:" in out.stdout


@pytest.mark.gpu
def test_dropin_multigpu_program_passes_on_one_gpu(tmp_path):
    """The C++ rank program of INTEGRATION.md section 5 (ExtractSift per image, then cusift_allgatherv_*) with
    world = 1: the shard still travels through RCCL (self send/recv inside one ncclGroup).  No torch in the process:
    RCCL is the system's, found next to the system HIP runtime."""
    build()
    out = subprocess.run([BIN_MULTI, "0", "1", str(tmp_path / "comm.id"), os.path.join(ROOT, "tests", "golden", "gray1.pgm"),
                          "3"], capture_output=True, text=True, timeout=300)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PASSED" in out.stdout and "rccl" in out.stdout.lower()


@pytest.mark.gpu
def test_scaling_bench_rank_program_prints_the_benchmark_line(tmp_path):
    """tests/cpp/scaling_bench: the multi-GPU benchmark step as a C++ rank program over the C ABI (no torch): W warm-up
    steps, K timed steps between two barriers, all-gatherv every step (trimmed records expanded on arrival), ONE JSON
    line from rank 0 with what the LIBRARY says about the communicator.  World 1 here (self send / recv through RCCL)."""
    import json

    build()
    out = subprocess.run([BIN_SCALING, "0", "1", str(tmp_path / "comm.id"), os.path.join(ROOT, "tests", "golden", "gray1.pgm"),
                          "4", "2", "5", "640", "480", "2"], capture_output=True, text=True, timeout=300)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["rccl_version"] > 20000 and "rccl" in d["config"]["rccl_library"]
    assert d["config"]["gather_record_bytes"] == 540 and len(d["config"]["ms_per_step_by_rank"]) == 1
    assert d["records_gathered_per_step"] == d["keypoints_per_step_rank0"] > 100


@pytest.mark.gpu
def test_dropin_pipeline_program_runs_on_gpu():
    """Consecutive batches rotated over several contexts through the C ABI alone (the throughput mode of bench.py,
    from C++): every context must deliver the same keypoint count for the same images."""
    import re

    build()
    pgm = os.path.join(ROOT, "tests", "golden", "gray1.pgm")
    counts = []
    for contexts in ("1", "3"):
        out = subprocess.run([BIN_PIPE, pgm, contexts, "6", "5", "640", "480"], capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        m = re.search(r"pipeline: (\d+) contexts, 6 batches of 5 images 640x480: ([0-9.]+) ms per batch, ([0-9.]+) Gpix/s, "
                      r"(\d+) keypoints per batch", out.stdout)
        assert m, out.stdout
        counts.append(int(m.group(4)))
    assert counts[0] == counts[1] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("threads,frames,w,h", [(4, 3, 640, 480), (3, 2, 1366, 768)])
def test_dropin_threads_program_equals_single_thread(threads, frames, w, h):
    """An unchanged cuSIFT program called from several host threads (each its own SiftData + cuImage, the reference's
    calls only): include/cuSIFT.h gives every calling thread its own implicit context, and every image's SiftData equals
    the single-thread run's bit for bit."""
    build()
    out = subprocess.run([BIN_THREADS, os.path.join(ROOT, "tests", "golden", "gray1.pgm"), str(threads), str(frames), "2",
                          str(w), str(h)], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert re.search(r"threads: %d threads x %d frames %dx%d: .* all equal" % (threads, frames, w, h), out.stdout), out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [2, 4])
def test_dropin_hostpipe_program_equals_blocking_extraction(depth):
    """cusift_pipe_* from plain C++: 8-bit frames in pinned host memory in, SiftData on the host out, `depth` batches in
    flight -- every image's keypoints equal those of the blocking cusift_extract_host on that frame alone."""
    build()
    out = subprocess.run([BIN_HOSTPIPE, os.path.join(ROOT, "tests", "golden", "gray1.pgm"), "7", "5", str(depth)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert re.search(r"hostpipe: 7 batches of 5 images 640x480, depth %d: .* all equal" % depth, out.stdout), out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,octaves,world", [
    (1024, 2048, 5, 1),    # one rank: the band entry points on the whole image, no communicator
    (1024, 2048, 7, 4),    # four ranks as four threads over the in-process transport; octaves 4..6 collapse onto rank 0
    (1000, 1531, 6, 3),    # uneven strips, ragged octave widths
])
def test_dropin_tiled_program_equals_whole_image(W, H, octaves, world):
    """The C++ tiled driver (cusift_tiled_* + cusift_allgatherv, plain g++ over include/cusift_amd.h): every rank's merged
    SiftData == the whole-image extraction, bit for bit."""
    from fake_transport import fake_rccl_path

    build()
    cmd = [BIN_TILED, os.path.join(ROOT, "tests", "golden", "gray1.pgm"), str(W), str(H), str(octaves), str(world)]
    if world > 1:
        cmd.append(fake_rccl_path())
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PASSED" in out.stdout and "identical" in out.stdout
