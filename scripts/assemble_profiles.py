"""Copies what scripts/profile_round.sh collected (gpurun_out/<dir>) into profiles/<round> — the files DESIGN.md §6 and
bench.py (roofline.traffic) cite — and prints the table rows.  usage: python scripts/assemble_profiles.py gpurun_out/round2 r02"""
import csv, glob, json, os, shutil, sys
src = sys.argv[1]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", sys.argv[2] if len(sys.argv) > 2 else "r02")
d = json.load(open(src + "/summary.json"))
traffic = {}
how = ("rocprofv3 --pmc, separate passes of `python bench.py --scene S --steps K --cpu-seconds 0` with the tuned schedule forced (PBR_PLAN): "
       "reads = 128*TCC_EA0_RDREQ_128B + 64*..._64B + 32*..._32B (= 2 x FETCH_SIZE KB on gfx950, calibrated with scripts/calibrate.py); "
       "writes = WRITE_SIZE KB; Infinity-Cache hits are included; the timed path-tracing launch only")
for sc, r in d.items():
    b = r["bench"]; cfg = b["config"]
    # the stats pass may have been redone (scripts/profile_stats_only.sh): read it from its files, not from the summary
    r["bench_under_rocprof"] = json.loads(open("%s/stats_%s.json" % (src, sc)).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(glob.glob("%s/stats_%s/*/*_kernel_stats.csv" % (src, sc))[0])))
    r["kernel_stats"] = [x for x in rows if "ptk::" in x["Name"]]
    os.makedirs(dst + "/" + sc, exist_ok=True)
    json.dump(b, open(dst + "/" + sc + "/bench_line.json", "w"), indent=1)
    json.dump(r["bench_under_rocprof"], open(dst + "/" + sc + "/bench_line_under_rocprof.json", "w"), indent=1)
    for name in ("kernel_stats", "domain_stats", "kernel_trace"):
        f = glob.glob("%s/stats_%s/*/*_%s.csv" % (src, sc, name))
        if f:
            shutil.copy(f[0], "%s/%s/%s.csv" % (dst, sc, name))
    samples = cfg["width"] * cfg["height"] * b["steps"]
    rd = r.get("fabric_read_bytes_per_launch", 0.0); wr = r.get("write_size_bytes", 0.0)
    traffic[sc] = {"width": cfg["width"], "height": cfg["height"], "max_depth": cfg["max_depth"], "brdf": cfg["brdf"], "steps": b["steps"],
                   "schedule": b.get("schedule"), "fabric_read_bytes_per_launch": rd, "fabric_write_bytes_per_launch": wr,
                   "bytes_per_sample": (rd + wr) / samples, "how": how}
    f = glob.glob("%s/stats_%s/*/*_kernel_trace.csv" % (src, sc))[0]
    kr = [x for x in csv.DictReader(open(f)) if "pathTracing" in x["Kernel_Name"]]
    kr.sort(key=lambda x: int(x["Start_Timestamp"]))
    last = kr[-1]; dur = (int(last["End_Timestamp"]) - int(last["Start_Timestamp"])) / 1e6
    u = r["bench_under_rocprof"]
    ks = [x for x in r.get("kernel_stats", []) if "pathTracing" in x["Name"]]
    avg = float(ks[0]["AverageNs"]) / 1e6 if ks else float("nan")
    calls = int(ks[0]["Calls"]) if ks else 0
    r["timed_launch_agreement"] = {"rocprof_timed_launch_ms": dur, "bench_launch_ms_same_run": u["roofline"]["launch_ms"], "kernel": last["Kernel_Name"][:60],
                                   "rocprof_stats_average_ms": avg, "rocprof_stats_calls": calls}
    print("         kernel_stats.csv: %d calls, average %.3f ms" % (calls, avg))
    p = r["pmc_timed_launch"]
    ns = [v for k, v in p.items() if k.startswith("duration_ns(SQ")][0]; cyc = ns * 2.4
    print("%-8s %7.1f %-12s launch %.3f ms  ach %.0f frac %.2f  traffic %.0f+%.0f  cpu %.2f (%.0fx)  VALU %.0f%% SALU %.0f%% lanes %.2f wait %.2f L2 %.3f  rocprof %.3f vs %.3f (%s)  read %.1f GB = %.2f TB/s" % (
        sc, b["value"], b["schedule"], b["roofline"]["launch_ms"], b["roofline"]["achieved"], b["roofline"]["frac"], rd / samples, wr / samples,
        b["cpu_baseline"]["value"], b["value"] / b["cpu_baseline"]["value"],
        100 * p["SQ_INSTS_VALU"] * 2.5 / 1024 / cyc, 100 * p["SQ_INSTS_SALU"] * 4.8 / 1024 / cyc, r["valu_lane_utilisation"], r["wave_wait_fraction"], r["l2_hit_rate"],
        dur, u["roofline"]["launch_ms"], u["schedule"], rd / 1e9, rd / ns / 1e3))
json.dump(d, open(dst + "/summary.json", "w"), indent=1)
json.dump(traffic, open(dst + "/pmc_traffic.json", "w"), indent=1)
