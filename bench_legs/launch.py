"""`python bench.py --gpus N` as a plain command: the parent starts the N ranks itself (children, before anything touches
a GPU) and relays rank 0's line; --dry-launch rehearses the ranks' rendezvous / shard / barrier / reduce on the CPU."""
import json
import os
import sys
import time

from .common import METRIC

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def spawn_ranks(n, argv):
    """One rank per GPU as child processes (torch.distributed.run on 127.0.0.1, a free port); rank 0's JSON line is
    passed through on stdout, everything else the children print goes to stderr.  Returns the launcher's exit code
    (non-zero if any rank failed).  Never an exec: this process stays the parent."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH_PY] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no JSON line\n")
        rc = 1
    return rc


def dry_launch(args):
    """What the ranks do around the timed region, without a GPU: rendezvous (gloo), shard, barrier, max-over-ranks."""
    import torch
    import torch.distributed as dist

    from cusift_amd.dist import shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(world * args.batch, rank, world)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0 + 1e-6 * rank], dtype=torch.float64)
    n_img = torch.tensor([hi - lo], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_img, op=dist.ReduceOp.SUM)
    tiled = rehearse_tiled_leg(rank, world) if world > 1 else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "tiled_leg": tiled,
                          "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_launch": True, "images_total": int(n_img.item()), "data": "none (launch rehearsal)"}),
              flush=True)
    return 0


class _RowBands:
    """The data-movement half of cusift_amd.tiling.StripExtractor on CPU tensors (the `_views` contract of
    tiling.exchange_halos), eight columns wide: every row holds its own GLOBAL row index, so a halo that arrived from the
    right neighbour at the right place can be told from one that did not."""

    def __init__(self, rank, world, plan):
        import torch

        self.rank, self.world, self.plan = rank, world, plan
        self.bands = []
        for o in range(plan.n_oct):
            lo, hi = plan.band(rank, o)
            a, b = plan.own(rank, o)
            t = torch.full((hi - lo, 8), -1.0, dtype=torch.float32)
            t[a - lo: b - lo] = torch.arange(a, b, dtype=torch.float32)[:, None]
            self.bands.append(t)

    def _views(self, o):
        pl = self.plan
        a, b = pl.own(self.rank, o)
        lo, hi = pl.band(self.rank, o)
        t, hal = self.bands[o], pl.halo
        return (t[a - lo: a - lo + hal] if self.rank > 0 else None,
                t[b - lo - hal: b - lo] if self.rank < self.world - 1 else None,
                t[0: a - lo] if self.rank > 0 else None,
                t[b - lo: hi - lo] if self.rank < self.world - 1 else None)


def rehearse_tiled_leg(rank, world):
    """The N > 1 leg of BASELINE configs[4] (bench_legs/configs.py: tiled_8192_all_ranks) without a GPU: the plan of the
    8192 x 8192 image over `world` ranks and its per-octave halo exchange over gloo (the pattern cusift_exchange_halos
    runs over RCCL), checked row by row; rank 0 returns what it saw plus the committed prediction for this many ranks."""
    import torch
    import torch.distributed as dist

    from cusift_amd.tiling import StripPlan, exchange_halos

    from .configs import TILED_H, TILED_W
    from .models import tiled_model

    plan = StripPlan(TILED_W, TILED_H, world, 5)
    bands = _RowBands(rank, world, plan)
    ok = True
    for o in range(plan.collapse):
        exchange_halos(bands, o)
        lo, hi = plan.band(rank, o)
        want = torch.arange(lo, hi, dtype=torch.float32)[:, None].expand(-1, 8)
        ok = ok and bool(torch.equal(bands.bands[o], want))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank != 0:
        return None
    model = tiled_model(TILED_W, TILED_H, 5, 0.70, 89469)["ranks"].get(str(world))
    return {"image": "%dx%d" % (TILED_W, TILED_H), "ranks": world, "tiled_octaves": plan.collapse,
            "collapse_octave": plan.collapse if plan.collapse < plan.n_oct else None,
            "rows_owned_octave0": [plan.own(k, 0)[1] - plan.own(k, 0)[0] for k in range(world)],
            "halo_rows": plan.halo, "halo_exchange_correct_on_every_rank": bool(flag.item() == 1),
            "predicted (whole image 0.70 ms, 89469 keypoints on one GPU: round 5's record)": model}
