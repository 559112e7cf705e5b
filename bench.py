#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (camera paths per second) of the path-tracing hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene cornell|sponza|dragon|hairball]

One "step" = one frame = one pass of the hot path over every pixel (SAMPLES = 1 path per pixel,
the reference's default).  The K timed steps run as ONE pbr_render call = one path-tracing launch
over all (pixel, frame) units + one foldFrames launch that applies the running mean in frame order
(several such pairs only if K frames x 16 B x pixels exceed 16 GiB), after W untimed warm-up
frames — during which the library also times its seven schedules on this scene and keeps the
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

Two opt-in modes of round 5 (include/pbr_hip.h, pbr_config): `--traversal six-order | eight-order` walks the same flat BVH
with every container's children ordered along the ray (same closest hits; fewer visits), `--arith native` computes with
gfx950's native sin / cos / rcp / sqrt / log / exp (statistical parity).  The default — the reference's one walk order and
the exact arithmetic, bit-identical to the oracle — is the headline; `config.traversal` / `config.arith` name the mode of a line.
`--force-dist` runs the collective leg with one rank (a real RCCL communicator and all-gather on one device).

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
# The L2's ceiling for THIS access shape, in requests per second: TCC_REQ of scripts/micro/gather_rate.hip's dependent random
# 32-B gathers from a 2 MiB (L2-resident) table at full occupancy / its duration, rocprofv3 --pmc TCC_REQ_sum
# (profiles/r04/experiments/l2_request_ceiling.txt).  Round 3 priced every TCC_REQ at 128 B against the guide's 18 TB/s of
# L2-resident row gathers and reported fractions of 1.07 and 1.19: TCC requests are not all 128 B.
L2_REQUEST_CEILING = 245e9
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

def measured_bound(fabric_frac, l2_frac, valu_busy):
    """What the counters of the profiled launch say binds the kernel (computed, not a table of
    strings): `fabric` when the traffic behind L2 reaches half the HBM peak, `l2-requests` when the L2 takes requests at
    90 % of its measured ceiling, `issue` when the vector ALU is 70 % busy, else `latency` (waves wait for dependent
    fetches with every one of those resources to spare)."""
    if fabric_frac is not None and fabric_frac >= 0.5:
        return "fabric"
    if l2_frac is not None and l2_frac >= 0.9:
        return "l2-requests"
    if valu_busy is not None and valu_busy >= 0.7:
        return "issue"
    return "latency"


def library_stamp():
    """Digest of the sources + flags the loaded csrc/libpbrhip.so was built from (build.py keeps it next to the library)."""
    try:
        with open(os.path.join(ROOT, "physically-based-rendering_amd", "csrc", "libpbrhip.so.srchash")) as f:
            return f.read().strip()
    except OSError:
        return None


def recorded_traffic(scene, w, h, depth, brdf, traversal=0, arith=0, schedule=None):
    """Fabric-side bytes PER SAMPLE from the committed PMC passes (profiles/rNN/pmc_traffic.json: rocprofv3 --pmc in
    separate runs; read = 128 x TCC_EA0_RDREQ_128B + 64 x _64B + 32 x _32B = 2 x FETCH_SIZE[KB] x 1024 on gfx950,
    write = WRITE_SIZE, calibrated with scripts/calibrate.py) of the newest round that profiled exactly this workload;
    None otherwise (PMC counters cannot be read from inside this run).  Every record carries the digest of the library
    it profiled (`srchash`) and the schedule: roofline_block refuses a record of another build or another schedule.
    Where the tuner's call between two schedules is close the round profiled both (keys <workload>_pN): the record of
    `schedule` is preferred, else the workload's first."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            records = json.load(open(path))
        except (OSError, ValueError):
            continue
        found = [(key, rec) for key, rec in records.items()      # keys: the scene, or scene_suffix for another size / mode / schedule of it
                 if rec.get("scene", key.split("_")[0]) == scene and (rec["width"], rec["height"], rec["max_depth"], rec["brdf"]) == (w, h, depth, brdf)
                 and (rec.get("traversal", 0), rec.get("arith", 0)) == (traversal, arith)]
        if found:
            # this run's schedule if it was profiled; else the schedule the tuner kept in the profiling run (the key without _pN)
            found.sort(key=lambda kr: (kr[1].get("schedule") != schedule, kr[0].rsplit("_p", 1)[-1].isdigit()))
            return dict(found[0][1], source=os.path.relpath(path, ROOT))
    return None


def roofline_block(scene, plan, traffic, algo_launch, per_launch_samples, kernel_s, stamp="unchecked", kernel=None):
    """The roofline of the dominant kernel, physical: `achieved` = bytes the memory system moved behind L2 for one launch
    (fabric reads + writes from the committed PMC passes of this workload — `traffic`, a record of
    profiles/rNN/pmc_traffic.json — Infinity-Cache hits included, scaled to this run's samples) / the launch's duration
    measured live (HIP events on the context's stream); `frac` = that / the HBM peak, <= 1 by construction.  The contract's
    algorithmic figure (every node visit priced at 32 B whether or not it left the CU) is `algorithmic_GBs`: it exceeds
    the peak on cache-resident scenes and ranks nothing.  `issue` and `l2` say what binds a scene whose working set
    lives in L2 / Infinity Cache."""
    roofline = {
        "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
        # the symbol of the kernel that ran, from the library (pbr_diag_last_kernel) — round 4 derived a name from the plan's
        # and printed pathTracingPhased for a launch of pathTracingDual
        "kernel": kernel, "launch_ms": kernel_s * 1e3,
        "algorithmic_bytes_per_launch": algo_launch, "algorithmic_GBs": algo_launch / kernel_s / 1e9,
        "bound_measured": None,      # no counters, no claim
    }
    if traffic is None:
        return roofline
    # counters of ANOTHER build of the kernels, or of another schedule, price nothing: say so instead of printing a
    # confident fraction (stamp = library_stamp() of the loaded library; "unchecked" only from tests of the arithmetic)
    stale = []
    if stamp != "unchecked" and traffic.get("srchash") != stamp:
        stale.append("profiled library %s, loaded library %s" % (str(traffic.get("srchash"))[:12], str(stamp)[:12]))
    if traffic.get("schedule") != plan:
        stale.append("profiled schedule %s, this run %s" % (traffic.get("schedule"), plan))
    if stale:
        roofline.update({"traffic_stale": True, "traffic_stale_why": "; ".join(stale), "traffic_source": traffic["source"]})
        return roofline
    roofline["traffic_stale"] = False
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
        l2_frac = req / kernel_s / L2_REQUEST_CEILING
        roofline["l2"] = {"requests_per_s": req / kernel_s, "peak": L2_REQUEST_CEILING, "frac": l2_frac,
                          # raw, never clamped: above 1 the measured ceiling, or the scaling of the profiled
                          # counters, is wrong for this workload, and that must stay visible
                          "exceeds_measured_ceiling": bool(l2_frac > 1.0),
                          "hit_rate": traffic.get("l2_hit_rate"),
                          # requests between L1 and L2 priced at a 128-B line each, over the algorithmic bytes:
                          # how many times the contract's traffic the L1s ask the L2s for (hairball, round 4: 2.2)
                          "l1_l2_amplification": req * 128.0 / algo_launch,
                          "note": "TCC_REQ per second against the request rate of dependent random 32-B gathers from an L2-resident table (scripts/micro/gather_rate.hip under rocprofv3 --pmc TCC_REQ_sum)"}
    verdict = measured_bound(roofline["frac"], roofline.get("l2", {}).get("frac"), roofline.get("issue", {}).get("valu_busy"))
    roofline["bound_measured"] = {
        "value": verdict, "fabric_frac": roofline["frac"], "l2_frac": roofline.get("l2", {}).get("frac"),
        "valu_busy": roofline.get("issue", {}).get("valu_busy"), "wave_wait_fraction": roofline.get("issue", {}).get("wave_wait_fraction"),
        "rule": "fabric if fabric_frac >= 0.5, else l2-requests if l2_frac >= 0.9, else issue if valu_busy >= 0.7, else latency",
    }
    # "hbm" only where it is true: the traffic behind L2 is what binds; otherwise the measured bound
    roofline["bound"] = "hbm" if verdict == "fabric" else verdict
    return roofline


def diff(a, b):
    return {k: a[k] - b[k] for k in a}


def cpu_baseline(pbr, scene, cfg, cam, px, budget_s):
    """The oracle (oracle/pt_oracle.c, kind 'port') on all host cores over a bounded sample of
    the same workload: whole frames of the same scene / resolution / depth until ~budget_s."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    # the timing build: -O3 -march=native -ffp-contract=off, compiled on this machine (the portable -O2 -mavx2
    # library the tests use stays the checker); falls back to the portable one where there is no compiler
    ref = oracle.Renderer(scene.desc, cfg, threads=cores, native=True)
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
        "sample": "%d frame(s) of the same %dx%d workload in %.1f s (oracle/pt_oracle.c, OpenMP, %d threads, %s)" % (
            frames, cfg.width, cfg.height, elapsed, cores,
            "gcc -O3 -march=native -ffp-contract=off, built on this host" if ref.native else "gcc -O2 -mavx2 -mfma -ffp-contract=off"),
    }


def elect_plan(votes):
    """The schedule most ranks' tuners settled on (-1 = a rank that has not settled does not vote); ties go to the
    plan the lowest rank voted for.  Every rank evaluates this on the same gathered list."""
    valid = [v for v in votes if v >= 0]
    if not valid:
        return -1
    return max(valid, key=lambda v: (valid.count(v), -valid.index(v)))


_PORT_HOLDERS = []     # launch_ranks: the rendezvous port stays reserved while the parent lives


def launch_ranks(n, argv, popen=subprocess.Popen, environ=None):
    """`--gpus N` without a launcher: start the N ranks ourselves — N copies of this script with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, the rendezvous on 127.0.0.1 — and return the child processes.  Runs in
    a parent that has not imported torch or the HIP library: a process that has initialised the GPU must neither fork
    nor exec.  Rank 0 inherits stdout (its one JSON line is the run's output); the other ranks' stdout goes to stderr."""
    environ = dict(os.environ if environ is None else environ)
    # The rendezvous port: bound here (SO_REUSEADDR, never listening) and KEPT bound for the parent's lifetime, so that
    # nothing else on the machine is handed the number; rank 0's store binds and listens on it with SO_REUSEADDR too,
    # which Linux allows while the other holder does not listen (checked with torch's TCPStore: it binds, accepts and
    # serves while this socket is open).  Round 3 bound, closed and handed the number on: a window of a process start.
    holder = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    holder.bind(("127.0.0.1", 0))
    port = holder.getsockname()[1]
    _PORT_HOLDERS.append(holder)
    procs = []
    for rank in range(n):
        env = dict(environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # dmabuf IPC.  Where this comes from: the build's environment notes for this GPU pool ("the host driver only
        # supports dmabuf IPC, and without it RCCL / CUDA-tensor sharing across processes fails with
        # hipIpcGetMemHandle: invalid argument"; the variable is exported on the pool's boxes already).  This build has
        # never had two devices to exercise it on: it is passed on, not verified.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # every rank in its own session = its own process group: end_ranks signals the GROUP (os.killpg), so the parent ends
        # a rank AND whatever it started; and a rank asks the kernel for SIGTERM when this parent dies (BENCH_PARENT_PID ->
        # die_with_parent: prctl PR_SET_PDEATHSIG), so that a parent killed outright (SIGKILL, OOM) leaves no orphans holding
        # the GPUs in a half-finished all-gather
        env["BENCH_PARENT_PID"] = str(os.getpid())
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                           stdout=None if rank == 0 else sys.stderr, start_new_session=True))
    return procs


def die_with_parent():
    """A self-launched rank: SIGTERM from the kernel when the launching parent dies, however it dies."""
    parent = os.environ.get("BENCH_PARENT_PID")
    if not parent:
        return
    try:
        import ctypes
        import signal
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG
    except (OSError, AttributeError):
        pass
    if os.getppid() != int(parent):     # it died between our start and the prctl
        sys.exit(143)


def signal_rank(p, sig):
    """Signal a rank's whole process group (it leads its own session); the bare PID where there is no group to address."""
    try:
        os.killpg(p.pid, sig)
    except (ProcessLookupError, PermissionError, OSError, AttributeError):
        try:
            p.send_signal(sig)
        except (ProcessLookupError, OSError):
            pass


def end_ranks(procs, grace_s=10.0):
    """Terminate the ranks that still run — each rank's own process group, by its exact id — wait, then kill what is left."""
    import signal
    for p in procs:
        if p.poll() is None:
            signal_rank(p, signal.SIGTERM)
    deadline = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            signal_rank(p, signal.SIGKILL)
    for p in procs:
        if p.poll() is None:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass


_STOP = {"signal": None}


def arm_signals():
    """SIGTERM / SIGINT handlers of the parent, installed BEFORE the ranks are started (a signal that arrives while they
    start is then seen by wait_ranks' first poll, not lost).  Returns the previous handlers."""
    import signal

    def on_signal(signum, _frame):
        _STOP["signal"] = signum

    previous = {}
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            previous[sig] = signal.signal(sig, on_signal)
        except ValueError:          # not the main thread (tests): no handlers, the rest works as before
            pass
    return previous


def wait_ranks(procs, poll_s=0.05, limit_s=None, previous=None):
    """Wait for every rank; the first one that fails ends the others (their exact PIDs), and its code is the run's.
    SIGTERM / SIGINT to this parent (a `timeout` around `python bench.py --gpus 8` signals only the parent) end the ranks
    before the parent exits — no orphans holding the GPUs in a half-finished all-gather —, and so does the wall-clock
    limit `limit_s`, after which the run fails with code 124."""
    import signal
    stop = _STOP
    if previous is None:
        previous = arm_signals()
    code, started = 0, time.monotonic()
    try:
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if failed:
                code = failed[0]
                end_ranks(procs)
                break
            if stop["signal"] is not None:
                sys.stderr.write("bench.py: signal %d — ending %d rank(s)\n" % (stop["signal"], sum(p.poll() is None for p in procs)))
                end_ranks(procs)
                code = 128 + stop["signal"]
                break
            if limit_s is not None and time.monotonic() - started > limit_s:
                sys.stderr.write("bench.py: the ranks did not finish within %.0f s — ending them\n" % limit_s)
                end_ranks(procs)
                code = 124
                break
            time.sleep(poll_s)
    finally:
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    for p in procs:
        if p.returncode not in (0, None) and code == 0:
            code = p.returncode
    return code


MODE_LEGS = (
    # (key in the line, traversal, arith, what the mode's parity claim is)
    ("eight-order", 2, 0, "tolerance"),            # same closest hits as the reference order except between faces at equal distance
    ("eight-order+native", 2, 1, "statistical"),   # native sin / cos / rcp / sqrt / log / exp: another sample of the same estimator
)


def mode_leg(pbr, scene, base_cfg, cam, px, args, device, depth, traversal, arith, parity, min_repeats=5):
    """One opt-in mode of the library on the headline's workload, same --steps: its own context, its own schedule tuning
    (untimed set-up, as for the headline), W warm-up frames, then the K-step render repeated at least `min_repeats` times —
    the median counts.  N = 1 only.  `roofline` is priced only from the mode's OWN committed PMC record (recorded_traffic
    matches traversal and arith; roofline_block refuses another build's or schedule's counters)."""
    cfg = type(base_cfg).from_buffer_copy(base_cfg)
    cfg.traversal, cfg.arith = traversal, arith
    w, h = int(cfg.width), int(cfg.height)
    dev = pbr.Device(device)
    try:
        dev.upload_scene(scene.desc)
        dev.configure(cfg)
        t_setup, setup_frames = time.perf_counter(), 0
        if args.plan >= 0:
            dev.pin_plan(args.plan)
        else:
            budget = dev.tune_budget()
            while setup_frames < budget or dev.last_plan()[1] < 0:
                n = max(1, min(args.steps, 4 * budget - setup_frames))
                dev.render(setup_frames, pbr.frame_seeds(setup_frames, n), px, cam)
                setup_frames += n
                if setup_frames >= 4 * budget:
                    break
        t_setup = time.perf_counter() - t_setup
        dev.reset_accum()
        if args.warmup > 0:
            dev.render(0, pbr.frame_seeds(0, args.warmup), px, cam)
        before, runs, first = dev.counters(), [], args.warmup
        while len(runs) < min_repeats or (sum(r[0] for r in runs) < 0.25 and len(runs) < 15):
            t0 = time.perf_counter()
            dev.render(first, pbr.frame_seeds(first, args.steps), px, cam)
            elapsed = time.perf_counter() - t0
            trace_ms, launches = dev.last_trace()
            runs.append((elapsed, trace_ms / launches / 1e3, launches))
            first += args.steps
        counters = {k: v / len(runs) for k, v in diff(dev.counters(), before).items()}
        elapsed, kernel_s, launches = sorted(runs)[len(runs) // 2]
        samples = w * h * args.steps * int(cfg.samples)
        assert counters["paths"] == samples, (counters, samples)
        algo = algorithmic_bytes(counters, w * h * args.steps)
        plan = dev.last_plan()[0]
        traffic = recorded_traffic(args.scene, w, h, depth, int(cfg.brdf), traversal, arith, schedule=plan)
        return {
            "value": samples / elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": elapsed * 1e3 / args.steps,
            "repeats": len(runs), "ms_per_step_all": [round(r[0] * 1e3 / args.steps, 5) for r in runs],
            "parity": parity, "traversal": ("reference", "six-order", "eight-order", "eight-order-compact")[traversal], "arith": ("exact", "native")[arith],
            "schedule": plan, "deal": dev.last_deal()[0], "setup_frames": setup_frames, "setup_s": round(t_setup, 3),
            "per_sample": {"node_visits": counters["nodes"] / samples, "triangle_tests": counters["tris"] / samples,
                           "shaded_hits": counters["hits"] / samples, "algorithmic_bytes": algo / samples},
            "scene_device_bytes": dev.scene_bytes(),
            "roofline": roofline_block(args.scene, plan, traffic, algo / launches, samples / launches, kernel_s, stamp=library_stamp(), kernel=dev.last_kernel()),
        }
    finally:
        dev.close()


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
    ap.add_argument("--plan", type=int, default=-1,
                    help="pin schedule 0..6 (refill-lean, refill-wide, phased-lean, phased-wide, phased-mid, refill-mid, phased-dual) instead of tuning; profiling runs")
    ap.add_argument("--traversal", default="reference", choices=["reference", "six-order", "eight-order", "eight-order-compact"],
                    help="pbr_config.traversal: the reference's one walk order (default, the headline), or the opt-in ray-ordered walk over the same flat BVH")
    ap.add_argument("--arith", default="exact", choices=["exact", "native"],
                    help="pbr_config.arith: every builtin one exact definition (default, the headline), or gfx950's native sin / cos / rcp / sqrt / log / exp")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: run the multi-GPU leg all the same — init_process_group(backend, world_size=1), the all-gather on pbr_export_tiles / pbr_import_tiles device pointers — and check the gathered frame")
    ap.add_argument("--modes", default="auto", choices=["auto", "off"],
                    help="auto: a one-GPU run of the default mode also times the opt-in modes (eight-order, eight-order + native) on the same workload and steps and reports them under \"modes\"; `value` stays the default mode")
    ap.add_argument("--dump", default="", help="rank 0 writes the gathered / rendered frame to this .npy")
    ap.add_argument("--rank-limit", type=float, default=1500.0, help="self-launched ranks (--gpus N without a launcher): wall-clock seconds after which the parent ends them and fails")
    ap.add_argument("--hold-seconds", type=float, default=3.0, help="N = 1: keep the GPU rendering (untimed) this long after the timed region, so that an outside utilisation sampler sees the GPU leg at all")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the parent of N ranks.  Nothing GPU-related has been imported yet.
        previous = arm_signals()
        sys.exit(wait_ranks(launch_ranks(args.gpus, sys.argv[1:]), limit_s=args.rank_limit, previous=previous))
    die_with_parent()
    if world != args.gpus:
        args.gpus = world

    # ONE line on stdout: RCCL prints a five-line version banner to the C-level stdout when a communicator is created
    # (seen with --force-dist on an MI355X), torch and the HIP runtime may print warnings there too.  From here on file
    # descriptor 1 is stderr; the JSON line goes to the descriptor the process was started with.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import pbr_loader
    pbr = pbr_loader.load()

    dist = torch = None
    multi = world > 1 or args.force_dist          # the collective leg runs (N = 1 with --force-dist: a one-rank RCCL communicator)
    if multi:
        import torch
        import torch.distributed as dist
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:      # --force-dist without a launcher: a rendezvous of one
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            holder = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            holder.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(holder.getsockname()[1])
            _PORT_HOLDERS.append(holder)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    cfg.traversal = {"reference": 0, "six-order": 1, "eight-order": 2, "eight-order-compact": 3}[args.traversal]
    cfg.arith = {"exact": 0, "native": 1}[args.arith]

    dev = pbr.Device(local_rank)
    dev.upload_scene(scene.desc)
    dev.configure(cfg)

    # The CPU baseline leg FIRST (rank 0 of a one-GPU run): the GPU leg then ends the process, and an outside utilisation
    # sampler that looks at the last seconds of the run sees the GPU at work, not 12 s of host cores (BENCH_r03: gpu_busy 0
    # on all samples — the 0.3 s GPU leg came first).  What a baseline has to time: PathTracer.cpp:43-71, one frame per call.
    baseline = None
    if world == 1 and rank == 0 and args.cpu_seconds > 0:
        baseline = cpu_baseline(pbr, scene, scene.config(w, h), cam, px, args.cpu_seconds)

    gather_out = gather_in = None
    if multi:
        n_floats = dev.tile_bytes() // 4
        gather_in = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
        gather_out = torch.zeros(n_floats * world, dtype=torch.float32, device="cuda")

    def sync():
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def gather():
        # every rank's accumulated tiles -> the full frame on every rank: the one collective of the path (RCCL all-gather)
        if not multi:
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
        budget = dev.tune_budget()                             # 206 frames at 1080p on one GPU (the upper bound: a close call between two plans is timed twice), N x as many on a rank of N (1/N of the pixels each)
        while setup_frames < budget or dev.last_plan()[1] < 0:
            n = max(1, min(args.steps, 4 * budget - setup_frames))
            dev.render(setup_frames, pbr.frame_seeds(setup_frames, n), px, cam)
            setup_frames += n
            if setup_frames >= 4 * budget:
                break
    plan_votes = None
    if multi and not forced:
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
        # the closing side: the all-gather IS the barrier (no rank has its frame before every rank has contributed its tiles)
        # and gather() ends in a device synchronize; a second dist.barrier() here only added its own latency to a
        # region of a few milliseconds.  The maximum over the ranks is taken below (all_reduce MAX of the per-run times).
        elapsed = time.perf_counter() - t0
        runs.append({"elapsed": elapsed, "render": t_render, "kernel_ms": kernel_ms, "trace_ms": trace_ms, "launches": trace_launches})
        first += args.steps
        more = (len(runs) < args.repeats) if args.repeats > 0 else (sum(r["elapsed"] for r in runs) < 0.25 and len(runs) < 15)
        if multi:                                      # every rank must take the same decision
            flag = torch.tensor([1 if more else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            more = bool(int(flag[0]))
        if not more:
            break
    plan, tuned = dev.last_plan()
    kernel_name = dev.last_kernel()
    fit = dev.launch_fit()
    repeats = len(runs)
    gathered_ok = None
    if args.force_dist and world == 1:
        # the gathered frame (all-gather of one rank's tiles, scattered by pbr_import_tiles) must be the rendered frame
        gathered_ok = bool(np.array_equal(dev.read_full(), dev.read_output(), equal_nan=True))
        if not gathered_ok:
            raise SystemExit("bench.py --force-dist: the frame behind the %s all-gather differs from the rendered one" % args.backend)

    counters = diff(dev.counters(), before)
    if rank == 0 and args.dump:
        np.save(args.dump, dev.read_full() if multi else dev.read_output())
    # the opt-in modes on the same workload and steps (N = 1, default-mode runs only): extra keys, `value` is untouched
    modes = None
    if world == 1 and not multi and args.modes == "auto" and (args.traversal, args.arith) == ("reference", "exact"):
        modes = {}
        for key, traversal, arith, parity in MODE_LEGS:
            if pbr.hip.pbr_mode_built(traversal, arith) != 1:
                modes[key] = {"value": None, "note": "this library was built without that mode"}
                continue
            modes[key] = mode_leg(pbr, scene, cfg, cam, px, args, local_rank, depth, traversal, arith, parity)
    # untimed: keep rendering for --hold-seconds, so that a sampler with a period of a second or two sees the GPU leg at all
    # (the timed region of the driver's command is 0.3 s)
    held_frames = 0
    if world == 1 and args.hold_seconds > 0:
        t_hold = time.perf_counter()
        while time.perf_counter() - t_hold < args.hold_seconds:
            dev.render(first + held_frames, pbr.frame_seeds(first + held_frames, args.steps), px, cam)
            held_frames += args.steps
    # per repetition: max over ranks of the elapsed time and of the average launch duration; then the median repetition
    per_run = [[r["elapsed"], r["trace_ms"] / r["launches"] / 1e3, r["render"], r["elapsed"] - r["render"]] for r in runs]
    totals = [float(counters["nodes"]), float(counters["tris"]), float(counters["hits"]), float(counters["paths"])]
    rank_ms = None
    if multi:
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
        traffic = recorded_traffic(args.scene, w, h, depth, int(cfg.brdf), int(cfg.traversal), int(cfg.arith), schedule=plan)
        roofline = roofline_block(args.scene, plan, traffic, algo_launch, samples / world / trace_launches, kernel_s, stamp=library_stamp(), kernel=kernel_name)
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
                # the two opt-in modes of round 5 (include/pbr_hip.h); "reference" / "exact" is the reference's behaviour and the headline
                "traversal": args.traversal, "arith": args.arith,
                # what the mode costs in node memory: the reference-order stream, and the ordered walk's six / eight streams
                "scene_device_bytes": dev.scene_bytes(),
            },
            "repeats": repeats, "ms_per_step_all": [round(r[0] * 1e3 / args.steps, 5) for r in per_run],
            "setup_frames": setup_frames, "setup_s": round(t_setup, 3),
            "kernel_ms": kernel_ms, "schedule": plan, "schedule_tuned": tuned >= 0, "trace_launches": trace_launches,
            "deal": dev.last_deal()[0],      # the order the queue dealt its tiles in: "spatial", or "cost-classes" (short launches)
            "per_sample": {
                "node_visits": counters["nodes"] / samples, "triangle_tests": counters["tris"] / samples,
                "shaded_hits": counters["hits"] / samples, "algorithmic_bytes": algo / samples,
            },
            "roofline": roofline,
        }
        if fit is not None:
            # what the schedule tuner measured on rank 0: a launch of n frames costs fixed + n x per_frame
            out["launch_fit_ms"] = {"fixed": round(fit[0], 4), "per_frame": round(fit[1], 5)}
        if rank_ms is not None:
            out["per_rank_ms"] = rank_ms
            # How far from linear a split of this render CAN be: every rank's launch carries the same fixed cost D (ramp-up
            # + the drain of its longest paths, independent of N) next to 1 / N of the work W, so N ranks reach
            # ( W + D ) / ( W + N D ) of N x one GPU.  W from the slowest rank's measured render R: W = N ( R - D ).
            if fit is not None and world > 1:
                # D is paid once per LAUNCH, and a render that is chunked (frame buffer cap, tuner) has trace_launches of them
                R, D = max(rank_ms["render"]), fit[0] * trace_launches
                W = world * max(R - D, 0.0)
                out["expected_linear_frac"] = round((W + D) / (W + world * D), 4) if W + world * D > 0 else None
                out["expected_linear_frac_note"] = "( W + D ) / ( W + N D ): D = rank 0's fitted fixed cost per launch x the render's %d launch(es), W = N ( slowest rank's render - D ); the all-gather is not in it" % trace_launches
        if gathered_ok is not None:
            out["force_dist"] = {"backend": args.backend, "world_size": world, "gathered_frame_equals_rendered": gathered_ok,
                                 "gather_ms": round(per_run[median][3] * 1e3, 3)}
        if plan_votes is not None:
            out["plan_votes"] = plan_votes
        if baseline is not None:
            out["cpu_baseline"] = baseline
        if modes is not None:
            out["modes"] = modes
        out["held_frames_untimed"] = held_frames
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())

    if multi:
        dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
