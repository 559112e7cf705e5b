"""Round measurements -> profiles/<round>.

  python scripts/assemble_profiles.py --collect <dir> <workload ...>   on the GPU box, at the end of scripts/profile_round.sh:
        reads the rocprofv3 CSVs under <dir> and writes <dir>/summary.json
  python scripts/assemble_profiles.py <dir> r03                         here: copies what DESIGN.md and bench.py cite into
        profiles/r03 (bench lines, kernel stats / trace CSVs, summary.json, pmc_traffic.json) and prints the table rows

Fabric reads = 128 x TCC_EA0_RDREQ_128B + 64 x _64B + 32 x _32B (= 2 x FETCH_SIZE[KB] x 1024 on gfx950, calibrated with
scripts/calibrate.py); writes = WRITE_SIZE[KB] x 1024; Infinity-Cache hits are included in both.  Every number is of the
TIMED path-tracing launch (the last dispatch of the kernel in a run)."""
import collections, csv, glob, json, os, shutil, sys

HOW = ("rocprofv3 --pmc, separate passes of `python3 bench.py --scene S --steps K --plan P --cpu-seconds 0` (the schedule the tuner kept, pinned): "
       "reads = 128*TCC_EA0_RDREQ_128B + 64*..._64B + 32*..._32B (= 2 x FETCH_SIZE KB on gfx950, calibrated with scripts/calibrate.py); "
       "writes = WRITE_SIZE KB; Infinity-Cache hits are included; the timed path-tracing launch only")


def newest(pattern):
    """rocprofv3 names its files by process id; a directory that has been merged twice holds two runs: take the later one."""
    found = sorted(glob.glob(pattern), key=os.path.getmtime)
    return found[-1:] 


def last_line_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


PLAN_NAMES = ["refill-lean", "refill-wide", "phased-lean", "phased-wide", "phased-mid", "refill-mid", "phased-dual"]


def library_stamp():
    """Digest of the sources + flags csrc/libpbrhip.so was built from (build.py writes it next to the library): the counters
    collected here are of THAT build, and bench.py prices a run with them only while it loads the same one."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        with open(os.path.join(root, "physically-based-rendering_amd", "csrc", "libpbrhip.so.srchash")) as f:
            return f.read().strip()
    except OSError:
        return None


def collect(out, loads):
    # a second pass over the same directory (more workloads, or one of them again) adds to what the first one found
    try:
        summary = json.load(open(out + "/summary.json"))
    except (OSError, ValueError):
        summary = {}
    for key in loads:
        rec = {"library_srchash": library_stamp()}
        try:
            rec["bench"] = last_line_json("%s/bench_%s.json" % (out, key))
        except Exception as e:
            rec["bench_error"] = str(e)
        try:
            rec["bench_under_rocprof"] = last_line_json("%s/stats_%s.json" % (out, key))
            rows = list(csv.DictReader(open(newest("%s/stats_%s/*/*_kernel_stats.csv" % (out, key))[0])))
            rec["kernel_stats"] = [r for r in rows if "ptk" in r["Name"]]
        except Exception as e:
            rec["stats_error"] = str(e)
        pmc = collections.OrderedDict()
        for f in sorted(sum((newest(d + "/*/*_counter_collection.csv") for d in sorted(glob.glob("%s/pmc*_%s" % (out, key)))), [])):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            meta = {}
            for r in csv.DictReader(open(f)):
                if "pathTracing" in r["Kernel_Name"]:
                    d = int(r["Dispatch_Id"])
                    per[d][r["Counter_Name"]] += float(r["Counter_Value"])
                    per[d]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    meta[d] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
            if per:
                timed = per[max(per)]          # the last pathTracing dispatch is the timed launch
                for k, v in timed.items():
                    pmc[k if k != "_ns" else "duration_ns(" + "+".join(sorted(c for c in timed if c != "_ns"))[:40] + ")"] = v
                rec["dispatch_metadata_as_rocprofv3_reports_it"] = meta[max(per)]
        rec["pmc_timed_launch"] = pmc
        if "TCC_EA0_RDREQ_128B_sum" in pmc:
            rec["fabric_read_bytes_per_launch"] = 128 * pmc["TCC_EA0_RDREQ_128B_sum"] + 64 * pmc.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * pmc.get("TCC_EA0_RDREQ_32B_sum", 0)
            rec["fetch_size_x2_bytes"] = 2 * 1024 * pmc.get("FETCH_SIZE", 0)
            rec["write_size_bytes"] = 1024 * pmc.get("WRITE_SIZE", 0)
            rec["l2_hit_rate"] = pmc.get("TCC_HIT_sum", 0) / max(1.0, pmc.get("TCC_HIT_sum", 0) + pmc.get("TCC_MISS_sum", 0))
        if "SQ_ACTIVE_INST_VALU" in pmc:
            rec["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (64 * pmc["SQ_INSTS_VALU"])
            rec["wave_wait_fraction"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
        summary[key] = rec
    json.dump(summary, open(out + "/summary.json", "w"), indent=1)
    for key, rec in summary.items():
        b = rec.get("bench", {})
        print(key, "Msamples/s %.1f" % b.get("value", 0), b.get("schedule"), "fabric read %.1f GB/launch" % (rec.get("fabric_read_bytes_per_launch", 0) / 1e9),
              "L2 hit %.3f" % rec.get("l2_hit_rate", 0), "lane util %.3f wait %.3f" % (rec.get("valu_lane_utilisation", 0), rec.get("wave_wait_fraction", 0)))


def assemble(src, round_name):
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", round_name)
    d = json.load(open(src + "/summary.json"))
    traffic = {}
    for key, r in d.items():
        b = r["bench"]; cfg = b["config"]
        os.makedirs(dst + "/" + key, exist_ok=True)
        json.dump(b, open(dst + "/" + key + "/bench_line.json", "w"), indent=1)
        json.dump(r["bench_under_rocprof"], open(dst + "/" + key + "/bench_line_under_rocprof.json", "w"), indent=1)
        for name in ("kernel_stats", "domain_stats", "kernel_trace"):
            f = newest("%s/stats_%s/*/*_%s.csv" % (src, key, name))
            if f:
                shutil.copy(f[0], "%s/%s/%s.csv" % (dst, key, name))
        samples = cfg["width"] * cfg["height"] * b["steps"]
        p = r["pmc_timed_launch"]
        rd = r.get("fabric_read_bytes_per_launch", 0.0); wr = r.get("write_size_bytes", 0.0)
        sq_keys = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS")
        traffic[key] = {"scene": cfg["scene"], "width": cfg["width"], "height": cfg["height"], "max_depth": cfg["max_depth"], "brdf": cfg["brdf"], "steps": b["steps"],
                        "traversal": {"reference": 0, "six-order": 1, "eight-order": 2, "eight-order-compact": 3}[cfg.get("traversal", "reference")], "arith": {"exact": 0, "native": 1}[cfg.get("arith", "exact")],
                        "schedule": b.get("schedule"), "plan": PLAN_NAMES.index(b["schedule"]) if b.get("schedule") in PLAN_NAMES else None,
                        "srchash": r.get("library_srchash"),
                        "fabric_read_bytes_per_launch": rd, "fabric_write_bytes_per_launch": wr,
                        "bytes_per_sample": (rd + wr) / samples, "l2_requests_per_launch": p.get("TCC_REQ_sum"), "l2_hit_rate": r.get("l2_hit_rate"),
                        "sq": {k: p[k] for k in sq_keys if k in p}, "how": HOW}
        f = newest("%s/stats_%s/*/*_kernel_trace.csv" % (src, key))[0]
        kr = [x for x in csv.DictReader(open(f)) if "pathTracing" in x["Kernel_Name"]]
        kr.sort(key=lambda x: int(x["Start_Timestamp"]))
        last = kr[-1]; dur = (int(last["End_Timestamp"]) - int(last["Start_Timestamp"])) / 1e6
        u = r["bench_under_rocprof"]
        ks = [x for x in r.get("kernel_stats", []) if "pathTracing" in x["Name"]]
        avg = float(ks[0]["AverageNs"]) / 1e6 if ks else float("nan")
        calls = int(ks[0]["Calls"]) if ks else 0
        r["timed_launch_agreement"] = {"rocprof_timed_launch_ms": dur, "bench_launch_ms_same_run": u["roofline"]["launch_ms"], "kernel": last["Kernel_Name"][:60],
                                       "rocprof_stats_average_ms": avg, "rocprof_stats_calls": calls}
        ns = [v for k, v in p.items() if k.startswith("duration_ns(") and "SQ_INSTS_VALU" in "".join(c for c in p if not c.startswith("duration")) and k.startswith("duration_ns(SQ_ACTIVE_INST_VALU")][0]
        busy = p["SQ_INSTS_VALU"] / (1024 * 1e9 * ns * 1e-9)          # wave-instructions x 2.4 cycles over 1024 SIMDs at 2.4 GHz
        r["issue"] = {"valu_busy": busy, "salu_busy": p["SQ_INSTS_SALU"] * 2.0 / (1024 * 1e9 * ns * 1e-9), "lane_utilisation": r["valu_lane_utilisation"],
                      "useful_lane_throughput_frac": busy * r["valu_lane_utilisation"]}
        rdns = [v for k, v in p.items() if k.startswith("duration_ns(TCC_EA0_RDREQ")][0]
        print("%-26s %7.1f Msamples/s %-12s launch %.3f ms | fabric %4.0f + %3.0f B/sample = %.2f TB/s read (%.0f %% of 8) | L2 hit %.3f, %.1f G requests/s | VALU busy %.0f %% x lanes %.0f %% = %.0f %% | waiting %.0f %% | cpu %.2f (%.0fx) | rocprof stats %d calls avg %.3f ms, timed %.3f vs events %.3f" % (
            key, b["value"], b["schedule"], b["roofline"]["launch_ms"], rd / samples, wr / samples, rd / rdns / 1e3, 100 * rd / rdns / 1e3 / 8.0,
            r["l2_hit_rate"], p.get("TCC_REQ_sum", 0) / rdns, 100 * busy, 100 * r["valu_lane_utilisation"], 100 * busy * r["valu_lane_utilisation"],
            100 * r["wave_wait_fraction"], b["cpu_baseline"]["value"], b["value"] / b["cpu_baseline"]["value"], calls, avg, dur, u["roofline"]["launch_ms"]))
    json.dump(d, open(dst + "/summary.json", "w"), indent=1)
    json.dump(traffic, open(dst + "/pmc_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "--collect":
        collect(sys.argv[2], sys.argv[3:])
    else:
        assemble(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r05")
