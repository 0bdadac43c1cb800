"""Constants, the SURVEY 8d byte model and the state every leg of bench.py shares."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
FP32_VALU_PEAK_TF = 157.3  # ibid. "Peak FP32 (vector)": 256 CUs x 4 SIMDs x 32 lanes x 2 flop (FMA) x 2.4 GHz
PRETEST_SKIP_HEADLINE = 0.73  # octave-0 wave-rows of the headline images the threshold pre-test skips (content leg)
ALL_LEGS = ("single", "repeat", "two_stage", "host", "host_in", "content", "initblur0", "ragged", "configs", "match", "cpu")
METRIC = "Mpix/s pyramid + keypoints/s end-to-end, 1920x1080 batch"
STAGE_KEYS = ("scale_down", "detect_multi", "describe_all", "laplace_multi", "find_points_multi", "compute_orientations",
              "extract_descriptors", "total")


def octave_dims(w, h, n_oct):
    dims = [(w, h)]
    for _ in range(1, n_oct):
        w, h = w // 2, h // 2
        if w < 1 or h < 1:
            break
        dims.append((w, h))
    return dims


def algorithmic_bytes(w, h, n_oct, n_img):
    """per octave, blur+DoG 32 B/px, downsample 4 B/px in + 4 B/px out, extrema 28 B/px."""
    dims = octave_dims(w, h, n_oct)
    blur = sum(32 * a * b for a, b in dims) * n_img
    find = sum(28 * a * b for a, b in dims) * n_img
    down = sum(4 * dims[i][0] * dims[i][1] + 4 * dims[i + 1][0] * dims[i + 1][1] for i in range(len(dims) - 1)) * n_img
    return blur, down, find


def usable_cpus():
    cores = os.cpu_count() or 1
    # the CPUs this process may actually use: the cgroup quota if there is one (a GPU box hands a 16-CPU share of its
    # 256 hardware threads to a job), else the affinity mask
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return cores, usable


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


class Run:
    """What the legs share: the arguments, the modules that need a GPU (imported once by bench.py), the extractors, the
    inputs, the timed region's results and `out`, the line being assembled.  Attributes are set by bench.py / timed.py."""

    def __init__(self, args):
        self.args = args
        self.out = None          # rank 0 only
        self.stage = None        # single leg: the one-stream stage table (HIP events), read by later legs
        self.stage_overlapped = None
        self.spans_ms_per_step = None

    # ---- helpers used by several legs ----
    def make_images(self, fn, seeds):
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            return np.stack(list(pool.map(fn, seeds)))

    @staticmethod
    def stage_table(st, steps):
        return {k: round(st[k][0] / steps, 4) for k in STAGE_KEYS}

    def run_single_stream(self, extractor, imgs, steps, warm=2):
        """`steps` extractions on one stream with per-launch HIP events; returns (ms per step, stage dict)."""
        torch = self.torch
        for _ in range(warm):
            extractor.extract(imgs)
        torch.cuda.synchronize()
        extractor.ctx.timing_enable(True)
        extractor.ctx.timing_reset()
        t1 = time.perf_counter()
        for _ in range(steps):
            extractor.extract(imgs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / steps * 1e3
        st = extractor.ctx.timing_read()
        extractor.ctx.timing_enable(False)
        return ms, st

    def run_pipelined(self, imgs, steps, warm=None, **param_overrides):
        """`steps` extractions rotated over the E streams exactly as in the timed region (no gather); returns ms/step.
        param_overrides are set on every extractor for the duration."""
        torch, exs, pipe, E = self.torch, self.exs, self.pipe, self.E
        warm = E if warm is None else warm
        saved = [{k: getattr(x.params, k) for k in param_overrides} for x in exs]
        for x in exs:
            x.params.concurrent_batches = E
            for k, v in param_overrides.items():
                setattr(x.params, k, v)
        try:
            for _ in range(warm):
                pipe.submit(imgs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                pipe.submit(imgs)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps * 1e3
        finally:
            for x, sv in zip(exs, saved):
                for k, v in sv.items():
                    setattr(x.params, k, v)

    def leg_guard(self, name):
        return _LegGuard(self, name)


class _LegGuard:
    """An extra leg never costs the line: an exception inside it is recorded under `leg_errors` and the extractors'
    parameters are put back to the timed region's."""

    def __init__(self, run, name):
        self.run, self.name = run, name

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None or not issubclass(et, Exception):
            return False
        R = self.run
        R.out.setdefault("leg_errors", {})[self.name] = "%s: %s" % (et.__name__, ev)
        print("bench.py: leg %s failed: %s: %s" % (self.name, et.__name__, ev), file=sys.stderr)
        try:
            R.torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            pass
        for x in R.exs:
            x.params.concurrent_batches = 1 if x is R.ex else R.E
            x.params.init_blur = R.args.init_blur
            x.params.fused_detect = 0 if R.args.two_stage else 1
        return True
