"""`python bench.py --gpus N` as a plain command: the parent must start the N ranks itself (children, before anything
touches a GPU) and relay rank 0's line.  --dry-launch rehearses exactly that path on the CPU over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          timeout=300, env=e)


def test_gpus_2_spawns_its_own_ranks_and_prints_one_line():
    out = run_bench("--gpus", "2", "--dry-launch", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout  # exactly ONE line on stdout: everything else went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_launch"] is True and d["images_total"] == 128 and d["steps"] == 3
    # the N > 1 leg of BASELINE configs[4] (one 8192 x 8192 image strip-tiled over the ranks that are up): plan + halo
    # exchange rehearsed over gloo, every rank's halos checked row by row
    t = d["tiled_leg"]
    assert t["ranks"] == 2 and t["halo_exchange_correct_on_every_rank"] is True and t["rows_owned_octave0"] == [4096, 4096]
    assert t["tiled_octaves"] == 5 and t["collapse_octave"] is None


def test_gpus_8_dry_launch():
    """The shape the driver's scaling run has (`python bench.py --gpus 8`): eight ranks rendezvous over gloo on the CPU,
    shard 512 images 64 per rank (BASELINE configs[3]), barrier, reduce a time, one line."""
    out = run_bench("--gpus", "8", "--dry-launch", "--steps", "2", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dry_launch"] is True and d["images_total"] == 512 and d["steps"] == 2
    # ... and BASELINE configs[4] over the same eight ranks: 1024 base rows each, octaves 0..4 tiled (64 rows >= the 48-row
    # halo in octave 4), halos exchanged with both neighbours in every octave and correct on every rank; the committed
    # prediction for eight ranks rides along
    t = d["tiled_leg"]
    assert t["ranks"] == 8 and t["rows_owned_octave0"] == [1024] * 8 and t["tiled_octaves"] == 5
    assert t["halo_exchange_correct_on_every_rank"] is True
    pred = [v for k, v in t.items() if k.startswith("predicted")][0]
    assert pred["exchanges_in_the_chain"] == 5 and pred["halo_bytes_per_neighbour_per_direction_by_octave"][0] == 48 * 8192 * 4


def test_under_torchrun_the_ranks_are_not_spawned_again():
    """The driver's other form: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` -- WORLD_SIZE is
    set, so bench.py must NOT start ranks of its own; rank 0 alone prints the line."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    assert json.loads(lines[0])["n_gpus"] == 2


def test_a_failing_rank_fails_the_command():
    # --gpus 2 but the children are told a batch the dry run rejects: make a rank exit non-zero through a bad flag
    out = run_bench("--gpus", "2", "--dry-launch", "--no-such-flag")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_the_parent_never_imports_torch_or_the_extension():
    """The spawning parent must not touch a GPU: it may not even import torch or load libcusift_amd.so."""
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--dry-launch']\n"
            "import subprocess\n"
            "real = subprocess.Popen\n"
            "class P(real):\n"
            "    def __init__(self, *a, **k):\n"
            "        assert 'torch' not in sys.modules and 'cusift_amd.capi' not in sys.modules, 'parent loaded torch'\n"
            "        print('SPAWN-OK', file=sys.stderr)\n"
            "        super().__init__(*a, **k)\n"
            "subprocess.Popen = P\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(ROOT, "bench.py"))
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e)
    assert "SPAWN-OK" in out.stderr and out.returncode == 0, out.stderr[-2000:]
