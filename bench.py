#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (camera paths per second) of the path-tracing hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene cornell|sponza|dragon|hairball]

One "step" = one frame = one pass of the hot path over every pixel (SAMPLES = 1 path per pixel,
the reference's default).  The K timed steps run as ONE pbr_render call = one path-tracing launch
over all (pixel, frame) units + one foldFrames launch that applies the running mean in frame order
(several such pairs only if K frames x 16 B x pixels exceed 16 GiB), after W untimed warm-up
frames — during which the library also times its six schedules on this scene and keeps the
fastest (2 frames each, then the best two or three again on 32 frames each: W >= 112, the default, settles it
before the timed region; with a smaller W the difference is rendered as untimed set-up before the warm-up).  Scene arrays and the
accumulated image are resident in HBM before the timed region starts.  Default workload = the configuration
BASELINE.json's metric and north_star are quoted on: configs[3], Sponza-class (260 k triangles), 1920x1080, depth 3
(--scene cornell | dragon | hairball select configs[1] / [2] / [4]).  When the K timed steps take less than 250 ms
the K-step render is repeated (each repetition bracketed by barrier + synchronize, continuing the accumulation) and
the MEDIAN repetition is reported — `repeats` and `ms_per_step_all` say so.

N > 1 (one rank per GPU; launched by torch.distributed.run — or by this script itself: `python bench.py --gpus N`
without RANK / WORLD_SIZE in the environment starts its N ranks as child processes before anything touches a GPU, waits
for them and forwards rank 0's line): 8x8-pixel tiles are dealt
round-robin to the ranks (along a row-rotated order, tiles.py: a rank never gets whole tile columns), the scene is replicated, no collective on the
data path; the timed region ends with one RCCL all-gather of the compact tile buffers
(W*H*16/N bytes per rank) and the scatter into the full frame.  The total work is fixed, so
this is strong scaling.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured float4 copy ~6290
HBM_ACHIEVABLE_GBS = 6300.0
L2_PEAK_GBS = 18000.0   # rows of an L2-resident table gathered by every workgroup: 16.8-18.8 TB/s (MI355X_MICROARCH.md)
NUM_SIMDS = 1024        # 256 CUs x 4

WORKLOADS = {
    # scene: (generator kind, seed, triangles, max_depth, BASELINE.json config it stands for)
    "cornell": ("cornell", 1, 0, 8, "configs[1]: Cornell box 1920x1080, 256 spp, depth 8"),
    "dragon": ("dragon", 1, 870000, 3, "configs[2]: Dragon-class (procedural, ~870k tris) 1920x1080, depth 3"),
    "sponza": ("sponza", 2, 260000, 3, "configs[3]: Sponza-class (procedural, ~260k tris) 1920x1080, depth 3"),
    "hairball": ("hairball", 3, 2000000, 3, "configs[4]: hairball (procedural, 2M tris), depth 3"),
}


def algorithmic_bytes(counters, pixel_frames, samples_per_frame=1):
    """32 B per node visit + 36 B per triangle test + 52 B per shaded hit
    + 32 B (accumulator read + write) per pixel and frame."""
    return 32 * counters["nodes"] + 36 * counters["tris"] + 52 * counters["hits"] + 32 * pixel_frames


GATHER_CEILING_LINES = 56e9   # random dependent 32-B gathers from a table beyond L2, 128-B lines per second (scripts/micro/gather_rate.hip: 55-57 G/s)

# what the counters say binds the kernel on each workload; "hbm" is the contract's ceiling for this path
MEASURED_BOUND = {"cornell": "vector issue (77 % busy at 47 % of the lanes)",
                  "sponza": "slowest lane's fetch per node phase (cache-resident: L2 hit 84 %, 62 % of wave-cycles waiting)",
                  "dragon": "fabric bandwidth (5.0 TB/s of reads = 63 % of the HBM peak, 80 % of a streaming copy's rate)",
                  "hairball": "L2 request rate (21 TB/s of 128-B lines, above the guide's L2-resident gather rate)"}


def recorded_traffic(scene, w, h, depth, brdf):
    """Fabric-side bytes PER SAMPLE from the committed PMC passes (profiles/rNN/pmc_traffic.json: rocprofv3 --pmc in
    separate runs; read = 128 x TCC_EA0_RDREQ_128B + 64 x _64B + 32 x _32B = 2 x FETCH_SIZE[KB] x 1024 on gfx950,
    write = WRITE_SIZE, calibrated with scripts/calibrate.py) of the newest round that profiled exactly this workload;
    None otherwise (PMC counters cannot be read from inside this run)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            records = json.load(open(path))
        except (OSError, ValueError):
            continue
        for key, rec in records.items():      # keys: the scene, or scene_suffix for another size of it ("hairball_4k")
            if rec.get("scene", key.split("_")[0]) == scene and (rec["width"], rec["height"], rec["max_depth"], rec["brdf"]) == (w, h, depth, brdf):
                return dict(rec, source=os.path.relpath(path, ROOT))
    return None


def roofline_block(scene, plan, traffic, algo_launch, per_launch_samples, kernel_s):
    """The roofline of the dominant kernel, physical: `achieved` = bytes the memory system moved behind L2 for one launch
    (fabric reads + writes from the committed PMC passes of this workload — `traffic`, a record of
    profiles/rNN/pmc_traffic.json — Infinity-Cache hits included, scaled to this run's samples) / the launch's duration
    measured live (HIP events on the context's stream); `frac` = that / the HBM peak, <= 1 by construction.  The contract's
    algorithmic figure (every node visit priced at 32 B whether or not it left the CU) is `algorithmic_GBs`: it exceeds
    the peak on cache-resident scenes and ranks nothing.  `issue` and `l2` say what binds a scene whose working set
    lives in L2 / Infinity Cache."""
    roofline = {
        "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
        "kernel": "ptk::pathTracingPhased" if plan.startswith("phased") else "ptk::pathTracing", "launch_ms": kernel_s * 1e3,
        "algorithmic_bytes_per_launch": algo_launch, "algorithmic_GBs": algo_launch / kernel_s / 1e9,
        "bound_measured": MEASURED_BOUND.get(scene),
    }
    if traffic is None:
        return roofline
    scale = per_launch_samples / (traffic["width"] * traffic["height"] * traffic["steps"])      # this launch's samples / the profiled launch's
    per_launch = traffic["bytes_per_sample"] * per_launch_samples
    lines = traffic["fabric_read_bytes_per_launch"] / 128.0 * scale
    roofline.update({
        "achieved": per_launch / kernel_s / 1e9, "frac": per_launch / kernel_s / 1e9 / HBM_PEAK_GBS,
        "traffic": per_launch,
        "traffic_source": traffic["source"] + " (separate rocprofv3 --pmc passes of this workload with schedule %s, scaled to this run's samples)" % traffic["schedule"],
        "achievable_frac": per_launch / kernel_s / 1e9 / HBM_ACHIEVABLE_GBS,
        "gather_ceiling": lines / kernel_s / GATHER_CEILING_LINES,
        "gather_ceiling_note": "128-B lines fetched per second / 56 G/s, this chip's rate of dependent random 32-B gathers beyond L2 (scripts/micro/gather_rate.hip)",
    })
    sq = traffic.get("sq")
    if sq:
        # vector issue: wave-instructions x 2.4 cycles each (scripts/micro/valu_rate.hip) over the chip's 1024 SIMDs
        # at 2.4 GHz = instructions / (1024 x 1e9 x t); useful lanes = thread-cycles / (64 x instructions)
        insts, threads = sq["SQ_INSTS_VALU"] * scale, sq["SQ_THREAD_CYCLES_VALU"] * scale
        busy = insts / (NUM_SIMDS * 1e9 * kernel_s)
        roofline["issue"] = {
            "valu_busy": busy, "lane_utilisation": threads / (64.0 * insts), "frac": busy * threads / (64.0 * insts),
            "wave_wait_fraction": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
            "note": "frac = share of the chip's vector lane-throughput doing useful work = SQ_THREAD_CYCLES_VALU x 2.4 cycles / (64 lanes x SIMD-cycles)",
        }
    if traffic.get("l2_requests_per_launch"):
        req = traffic["l2_requests_per_launch"] * scale
        roofline["l2"] = {"request_GBs": req * 128.0 / kernel_s / 1e9, "peak": L2_PEAK_GBS, "frac": req * 128.0 / kernel_s / 1e9 / L2_PEAK_GBS,
                          "hit_rate": traffic.get("l2_hit_rate"), "note": "TCC_REQ x 128 B against the 17-19 TB/s the guide measures for L2-resident gathers"}
    return roofline


def diff(a, b):
    return {k: a[k] - b[k] for k in a}


def cpu_baseline(pbr, scene, cfg, cam, px, budget_s):
    """The oracle (oracle/pt_oracle.c, kind 'port') on all host cores over a bounded sample of
    the same workload: whole frames of the same scene / resolution / depth until ~budget_s."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    ref = oracle.Renderer(scene.desc, cfg, threads=cores)
    frames, t0, elapsed = 0, time.perf_counter(), 0.0
    # a band of rows per call keeps each call short; whole frames are what is reported
    while True:
        seed = float(pbr.frame_seeds(frames, 1)[0])
        weight = float(np.float32(frames) / np.float32(frames + 1))
        ref.image = ref.render_frame(seed, weight, px, cam)
        frames += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= budget_s or frames >= 64:
            break
    samples = cfg.width * cfg.height * frames
    return {
        "value": samples / elapsed / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "%d frame(s) of the same %dx%d workload in %.1f s (oracle/pt_oracle.c, OpenMP, %d threads)" % (
            frames, cfg.width, cfg.height, elapsed, cores),
    }


def elect_plan(votes):
    """The schedule most ranks' tuners settled on (-1 = a rank that has not settled does not vote); ties go to the
    plan the lowest rank voted for.  Every rank evaluates this on the same gathered list."""
    valid = [v for v in votes if v >= 0]
    if not valid:
        return -1
    return max(valid, key=lambda v: (valid.count(v), -valid.index(v)))


def launch_ranks(n, argv, popen=subprocess.Popen, environ=None):
    """`--gpus N` without a launcher: start the N ranks ourselves — N copies of this script with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, the rendezvous on 127.0.0.1 — and return the child processes.  Runs in
    a parent that has not imported torch or the HIP library: a process that has initialised the GPU must neither fork
    nor exec.  Rank 0 inherits stdout (its one JSON line is the run's output); the other ranks' stdout goes to stderr."""
    environ = dict(os.environ if environ is None else environ)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's only working transport on this pool
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                           stdout=None if rank == 0 else sys.stderr))
    return procs


def wait_ranks(procs, poll_s=0.05):
    """Wait for every rank; the first one that fails ends the others (their exact PIDs), and its code is the run's."""
    code = 0
    while any(p.poll() is None for p in procs):
        failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if failed:
            code = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(poll_s)
    for p in procs:
        if p.returncode not in (0, None) and code == 0:
            code = p.returncode
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=112)
    ap.add_argument("--scene", default="sponza", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--triangles", type=int, default=-1, help="override the scene's triangle budget")
    ap.add_argument("--depth", type=int, default=-1, help="override render.max_depth")
    ap.add_argument("--brdf", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device: rehearse the N > 1 path on a single GPU (tests); the gather then goes through host memory")
    ap.add_argument("--one-device", action="store_true", help="every rank uses device 0 (rehearsal only)")
    ap.add_argument("--repeats", type=int, default=0, help="repetitions of the K-step render (0 = until 250 ms have been timed, at most 15); the median is reported")
    ap.add_argument("--plan", type=int, default=int(os.environ.get("PBR_PLAN", "-1")),
                    help="pin schedule 0..5 (refill-lean, refill-wide, phased-lean, phased-wide, phased-mid, refill-mid) instead of tuning; profiling runs")
    ap.add_argument("--dump", default="", help="rank 0 writes the gathered / rendered frame to this .npy")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the parent of N ranks.  Nothing GPU-related has been imported yet.
        sys.exit(wait_ranks(launch_ranks(args.gpus, sys.argv[1:])))
    if world != args.gpus:
        args.gpus = world

    import pbr_loader
    pbr = pbr_loader.load()

    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    kind, seed, triangles, depth, label = WORKLOADS[args.scene]
    triangles = args.triangles if args.triangles >= 0 else triangles
    depth = args.depth if args.depth > 0 else depth
    pbr.cfg_reset()
    pbr.cfg_set(**{"render.max_depth": depth, "render.brdf": args.brdf})
    t_build = time.perf_counter()
    scene = pbr.HostScene.generate(kind, seed, triangles)
    t_build = time.perf_counter() - t_build

    w, h = args.width, args.height
    cfg, cam, px = scene.config(w, h), scene.camera(), pbr.pixel_dimension(w, h)
    cfg.tile_world, cfg.tile_rank = world, rank

    dev = pbr.Device(local_rank)
    dev.upload_scene(scene.desc)
    dev.configure(cfg)

    gather_out = gather_in = None
    if world > 1:
        n_floats = dev.tile_bytes() // 4
        gather_in = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
        gather_out = torch.zeros(n_floats * world, dtype=torch.float32, device="cuda")

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def gather():
        # every rank's accumulated tiles -> the full frame on every rank: the one collective of the path (RCCL all-gather)
        if world == 1:
            return
        dev.export_tiles(gather_in.data_ptr())
        if args.backend == "nccl":
            dist.all_gather_into_tensor(gather_out, gather_in)
        else:
            host_out = torch.empty(gather_out.numel(), dtype=torch.float32)
            dist.all_gather_into_tensor(host_out, gather_in.cpu())
            gather_out.copy_(host_out)
        torch.cuda.synchronize()
        dev.import_tiles(gather_out.data_ptr())

    # set-up, untimed and outside the W warm-up steps: the schedule tuner needs dev.tune_budget() frames of this scene +
    # configuration once (the counterpart of the reference's per-scene clBuildProgram), rendered in
    # calls of the length that will be timed, so that it settles on the plan that is fastest for THAT length.
    # N > 1: every rank tunes on its own share AT THE SAME TIME (the set-up takes as long as on one GPU, whatever N),
    # then the ranks vote — the plan most ranks settled on is pinned on all of them, so that all run one schedule and
    # none is a straggler of the closing all-gather because its own timing noise picked another plan.
    forced = args.plan >= 0
    if forced:
        dev.pin_plan(args.plan)
    setup_frames, t_setup = 0, time.perf_counter()
    if not forced:
        budget = dev.tune_budget()                             # 108 frames at 1080p on one GPU, N x as many on a rank of N (1/N of the pixels each)
        while setup_frames < budget or dev.last_plan()[1] < 0:
            n = max(1, min(args.steps, 4 * budget - setup_frames))
            dev.render(setup_frames, pbr.frame_seeds(setup_frames, n), px, cam)
            setup_frames += n
            if setup_frames >= 4 * budget:
                break
    plan_votes = None
    if world > 1 and not forced:
        where = "cuda" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([dev.last_plan()[1]], dtype=torch.int32, device=where)
        votes = torch.zeros(world, dtype=torch.int32, device=where)
        dist.all_gather_into_tensor(votes, mine)
        plan_votes = [int(v) for v in votes.cpu()]
        dev.pin_plan(elect_plan(plan_votes))
    t_setup = time.perf_counter() - t_setup
    dev.reset_accum()

    # warm-up: W frames (their own launch), accumulated image stays on the device
    if args.warmup > 0:
        dev.render(0, pbr.frame_seeds(0, args.warmup), px, cam)
    gather()          # RCCL connects its rings on first use: not a cost of the timed steps
    before = dev.counters()

    # the timed region: EXACTLY K steps between barrier + synchronize on both sides; repeated (continuing the
    # accumulation) while the repetitions so far took less than 250 ms, at most 15 times; the median repetition counts
    runs, first = [], args.warmup
    while True:
        sync()
        t0 = time.perf_counter()
        dev.render(first, pbr.frame_seeds(first, args.steps), px, cam)   # synchronous: returns after the launch completed
        t_render = time.perf_counter() - t0
        kernel_ms = dev.last_kernel_ms()            # the whole render: path tracing + foldFrames
        trace_ms, trace_launches = dev.last_trace()  # the path-tracing launches alone
        gather()
        sync()
        elapsed = time.perf_counter() - t0
        runs.append({"elapsed": elapsed, "render": t_render, "kernel_ms": kernel_ms, "trace_ms": trace_ms, "launches": trace_launches})
        first += args.steps
        more = (len(runs) < args.repeats) if args.repeats > 0 else (sum(r["elapsed"] for r in runs) < 0.25 and len(runs) < 15)
        if world > 1:                                  # every rank must take the same decision
            flag = torch.tensor([1 if more else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            more = bool(int(flag[0]))
        if not more:
            break
    plan, tuned = dev.last_plan()
    repeats = len(runs)

    counters = diff(dev.counters(), before)
    # per repetition: max over ranks of the elapsed time and of the average launch duration; then the median repetition
    per_run = [[r["elapsed"], r["trace_ms"] / r["launches"] / 1e3, r["render"], r["elapsed"] - r["render"]] for r in runs]
    totals = [float(counters["nodes"]), float(counters["tris"]), float(counters["hits"]), float(counters["paths"])]
    rank_ms = None
    if world > 1:
        where = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor(per_run, dtype=torch.float64, device=where)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mine = torch.zeros(world, 2, dtype=torch.float64, device=where)
        mid = sorted(range(repeats), key=lambda i: float(tmax[i, 0]))[repeats // 2]
        mine[rank, 0], mine[rank, 1] = per_run[mid][2] * 1e3, per_run[mid][3] * 1e3
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        rank_ms = {"render": [round(float(v), 3) for v in mine[:, 0]], "gather": [round(float(v), 3) for v in mine[:, 1]]}
        c = torch.tensor(totals, dtype=torch.float64, device=where)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counters = {"nodes": int(c[0]), "tris": int(c[1]), "hits": int(c[2]), "paths": int(c[3])}
        per_run = [[float(v) for v in row] for row in tmax]
    order = sorted(range(repeats), key=lambda i: per_run[i][0])
    median = order[repeats // 2]
    elapsed, kernel_s = per_run[median][0], per_run[median][1]
    kernel_ms, trace_launches = runs[median]["kernel_ms"], runs[median]["launches"]

    if rank == 0:
        samples = w * h * args.steps * int(cfg.samples)
        assert counters["paths"] == samples * repeats, (counters, samples, repeats)
        counters = {k: v / repeats for k, v in counters.items()}       # per K-step render (every repetition does the same work up to the seeds)
        algo = algorithmic_bytes(counters, w * h * args.steps)
        # per launch of the dominant kernel (the path-tracing kernel the auto-tuner settled on): each rank runs
        # trace_launches of them per render; the slowest rank's average launch duration
        algo_launch = algo / world / trace_launches          # SURVEY 8(d)'s per-sample figure x the samples one launch processes
        traffic = recorded_traffic(args.scene, w, h, depth, int(cfg.brdf))
        roofline = roofline_block(args.scene, plan, traffic, algo_launch, samples / world / trace_launches, kernel_s)
        out = {
            "metric": "Msamples/s (paths/s) @1080p fixed seed; 1/2/4/8 MI355X scaling",
            "value": samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": label if (w, h) == (1920, 1080) else label + " at %dx%d" % (w, h),
                "scene": args.scene, "triangles": scene.info["faces"], "bvh_nodes": scene.info["flat_nodes"],
                "width": w, "height": h, "spp": args.steps * int(cfg.samples), "max_depth": depth,
                "max_added_depth": int(cfg.max_added_depth), "brdf": int(cfg.brdf),
                "seeds": "seed_k = 0.0333 * (k + 1)", "tiles": "8x8 px, dealt round-robin to %d rank(s) along rows rotated by 5 * row columns" % world,
                "host_bvh_build_s": round(t_build, 3),
            },
            "repeats": repeats, "ms_per_step_all": [round(r[0] * 1e3 / args.steps, 5) for r in per_run],
            "setup_frames": setup_frames, "setup_s": round(t_setup, 3),
            "kernel_ms": kernel_ms, "schedule": plan, "schedule_tuned": tuned >= 0, "trace_launches": trace_launches,
            "per_sample": {
                "node_visits": counters["nodes"] / samples, "triangle_tests": counters["tris"] / samples,
                "shaded_hits": counters["hits"] / samples, "algorithmic_bytes": algo / samples,
            },
            "roofline": roofline,
        }
        if rank_ms is not None:
            out["per_rank_ms"] = rank_ms
        if plan_votes is not None:
            out["plan_votes"] = plan_votes
        if world == 1 and args.cpu_seconds > 0:
            cfg1 = scene.config(w, h)
            out["cpu_baseline"] = cpu_baseline(pbr, scene, cfg1, cam, px, args.cpu_seconds)
        if args.dump:
            np.save(args.dump, dev.read_full() if world > 1 else dev.read_output())
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
