"""The N > 1 path on CPU: two processes, gloo.  Each rank renders ONLY its tiles (with the
oracle standing in for the device — this is a test), packs them in the compact tile-major layout,
all-gathers, de-interleaves; the result must be bit-identical to the single-process frame
(SURVEY.md §8e invariant).  Also checks the layout helpers against each other."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, same_values

W, H, FRAMES = 64, 48, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _render_rows(pbr, orc, world, rank):
    """Full-frame oracle render restricted to the pixels of `rank`'s tiles."""
    pbr.cfg_reset()
    sc = pbr.HostScene.generate("cornell")
    cfg, cam, px = sc.config(W, H), sc.camera(), pbr.pixel_dimension(W, H)
    img = orc.Renderer(sc.desc, cfg).render(0, pbr.frame_seeds(0, FRAMES), px, cam)
    mask = pbr.tiles.rows_of_rank(W, H, world, rank)
    return np.where(mask[..., None], img, 0).astype(np.float32), img


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import pbr_loader
    pbr = pbr_loader.load()
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine, _ = _render_rows(pbr, orc, world, rank)
        local = torch.from_numpy(pbr.tiles.pack_rank_tiles(mine, world, rank).copy()).reshape(-1)
        gathered = torch.empty(world * local.numel(), dtype=local.dtype)
        dist.all_gather_into_tensor(gathered, local)
        frame = pbr.tiles.unpack_gathered(gathered.numpy(), W, H, world)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), frame)
        # bench.py's control traffic at N > 1: rank 0's tuned plan to everybody, and "one more repetition?" from rank 0
        choice = torch.tensor([4 if rank == 0 else -1], dtype=torch.int32)
        dist.broadcast(choice, src=0)
        more = torch.tensor([1 if rank == 0 else 0], dtype=torch.int32)
        dist.broadcast(more, src=0)
        # ... and its statistics: max over ranks of the elapsed time, per-rank slots summed into one table
        slow = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(slow, op=dist.ReduceOp.MAX)
        table = torch.zeros(world, dtype=torch.float64)
        table[rank] = 10.0 + rank
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
        np.save(os.path.join(out_dir, "ctl%d.npy" % rank), np.array([float(choice[0]), float(more[0]), float(slow[0])] + table.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_is_bit_identical_to_one_rank(tmp_path, pbr, oracle, world):
    """world = 8: the node the metric is quoted on (48 tiles: 6 per rank)."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    _, full = _render_rows(pbr, oracle, 1, 0)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % rank))
        assert same_values(got, full), "rank %d" % rank
        ctl = np.load(os.path.join(str(tmp_path), "ctl%d.npy" % rank))
        assert ctl.tolist() == [4.0, 1.0, float(world)] + [10.0 + r for r in range(world)]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_tile_layout_round_trip(pbr, world):
    rng = np.random.default_rng(world)
    w, h = 72, 40                                   # 45 tiles: not divisible by 2, 8
    img = rng.random((h, w, 4), dtype=np.float32)
    t = pbr.tiles
    assert same_values(t.from_tile_major(t.to_tile_major(img), w, h), img)
    bufs = np.stack([t.pack_rank_tiles(img, world, r) for r in range(world)])
    assert bufs.shape[1] == t.tile_counts(w, h, world)[3]
    assert same_values(t.unpack_gathered(bufs, w, h, world), img)
    masks = sum(t.rows_of_rank(w, h, world, r).astype(int) for r in range(world))
    assert (masks == 1).all()                       # every pixel has exactly one owner
    ids = np.concatenate([t.local_tile_ids(w, h, world, r) for r in range(world)])
    assert sorted(ids.tolist()) == list(range(45))


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 8), (1920, 1080, 4), (3840, 2160, 8), (72, 40, 3)])
def test_dealing_order_spreads_every_rank_over_columns_and_rows(pbr, w, h, world):
    """tiles.py / pt_kernel.hpp dealPositionOfTile: with plain row-major dealing a rank owns whole tile columns whenever
    tiles_x is a multiple of world (240 columns, 8 ranks at 1080p) — measured 2.6 % more work on the heaviest rank.
    Along the row-rotated order every rank has (almost) the same number of tiles in every column and in every row."""
    t = pbr.tiles
    tiles_x, tiles_y, total, per_rank = t.tile_counts(w, h, world)
    seen = np.zeros(total, int)
    for rank in range(world):
        ids = t.local_tile_ids(w, h, world, rank)
        seen[ids] += 1
        cols = np.bincount(ids % tiles_x, minlength=tiles_x)
        rows = np.bincount(ids // tiles_x, minlength=tiles_y)
        assert cols.max() - cols.min() <= 2, (rank, cols.min(), cols.max())
        assert rows.max() - rows.min() <= 2, (rank, rows.min(), rows.max())
    assert (seen == 1).all()


# ----------------------------------------------------------------------------------------------
# bench.py --gpus N without a launcher: it starts its own ranks
# ----------------------------------------------------------------------------------------------

def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # importing bench.py imports neither torch nor the HIP library
    return mod


def test_bench_launches_one_rank_per_gpu_with_a_local_rendezvous():
    bench = _bench_module()
    started = []

    class Fake:
        def __init__(self, cmd, env=None, stdout=None):
            started.append((cmd, env, stdout))

    procs = bench.launch_ranks(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], popen=Fake, environ={"PATH": os.environ["PATH"]})
    assert len(procs) == len(started) == 8
    ports = set()
    for rank, (cmd, env, stdout) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[2:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"]) == (str(rank), str(rank), "8")
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
        assert (stdout is None) == (rank == 0)          # only rank 0 writes to the run's stdout
    assert len(ports) == 1 and 0 < int(ports.pop()) < 65536


def test_bench_plan_election():
    bench = _bench_module()
    assert bench.elect_plan([4, 4, 5, 4, 5, 4, 4, 2]) == 4
    assert bench.elect_plan([5, 4]) == 5                 # tie: the lowest rank's vote
    assert bench.elect_plan([-1, 2, -1]) == 2            # ranks that have not settled do not vote
    assert bench.elect_plan([-1, -1]) == -1


def test_bench_first_failing_rank_ends_the_run():
    import subprocess
    bench = _bench_module()
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)"]),
             subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(7)"])]
    assert bench.wait_ranks(procs) == 7
    assert all(p.poll() is not None for p in procs)


def test_bench_gpus_8_without_a_launcher_spawns_instead_of_refusing():
    """`python bench.py --gpus 8` with no WORLD_SIZE must not stop at argument parsing (round 2 did: "needs a
    torch.distributed.run launch").  Here there is no GPU, so the eight ranks it starts fail loudly at their first device call —
    the product has no CPU fallback — and the parent reports that failure."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                          "--scene", "cornell", "--width", "64", "--height", "48", "--cpu-seconds", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run (covered by the gpu suite)")
    assert "torch.distributed.run launch" not in run.stderr
    # the ranks were started and refused to run without a device (torch.cuda.set_device or pbr_create, whichever comes first)
    assert run.returncode != 0 and ("No HIP GPUs" in run.stderr or "PbrError" in run.stderr)


def test_bench_roofline_is_physical_for_every_profiled_workload():
    """bench.py's roofline block from the committed PMC passes (profiles/r03/pmc_traffic.json) at each workload's own
    launch time (profiles/r03/summary.json): `frac` is fabric traffic / time / 8 TB/s — between 0 and 1 —, the issue-side
    and L2 fractions are below 1 as well, and the contract's algorithmic figure is reported beside them, not as `frac`."""
    import json
    bench = _bench_module()
    summary = json.load(open(os.path.join(ROOT, "profiles", "r03", "summary.json")))
    seen = 0
    for key, rec in summary.items():
        line = rec["bench"]
        cfg = line["config"]
        traffic = bench.recorded_traffic(cfg["scene"], cfg["width"], cfg["height"], cfg["max_depth"], cfg["brdf"])
        assert traffic is not None and traffic["source"].startswith(os.path.join("profiles", "r03")), key
        samples = cfg["width"] * cfg["height"] * line["steps"]
        seconds = line["roofline"]["launch_ms"] / 1e3
        block = bench.roofline_block(cfg["scene"], line["schedule"], traffic, line["per_sample"]["algorithmic_bytes"] * samples, samples, seconds)
        want = (traffic["fabric_read_bytes_per_launch"] + traffic["fabric_write_bytes_per_launch"]) / seconds / 8e12
        assert abs(block["frac"] - want) < 1e-9 and 0.0 < block["frac"] <= 1.0, (key, block["frac"])
        assert block["achieved"] <= block["peak"] and block["bound"] == "hbm" and block["unit"] == "GB/s"
        assert 0.0 < block["issue"]["frac"] < 1.0 and 0.0 < block["issue"]["lane_utilisation"] < 1.0
        assert block["algorithmic_GBs"] > 0 and "algorithmic_bytes_per_launch" in block
        seen += 1
    assert seen == 5                                   # cornell, sponza, dragon, hairball, hairball at 3840 x 2160
    # a workload nobody profiled: no invented number
    none = bench.roofline_block("sponza", "phased-mid", None, 1e9, 1e6, 1e-3)
    assert none["frac"] is None and none["traffic"] is None and none["algorithmic_GBs"] == 1e9 / 1e-3 / 1e9
