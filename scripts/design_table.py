"""The measurement table of DESIGN.md section 6 from profiles/<round>/summary.json + pmc_traffic.json (written by
scripts/assemble_profiles.py): one row per profiled workload.  `python scripts/design_table.py r05 [--write]` prints the
rows; --write replaces the block between the r05-table markers in DESIGN.md."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
d = json.load(open(os.path.join(ROOT, "profiles", rnd, "summary.json")))
t = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")))
LABEL = {"cornell": "Cornell-class, depth 8 (`configs[1]`)", "dragon": "Dragon-class (`configs[2]`)", "sponza": "Sponza-class (`configs[3]`, **the bench default**)",
         "hairball": "hairball, 1080p", "hairball_4k": "hairball at **3840×2160** (`configs[4]`'s own size)"}
L2_CEILING = 245e9
rows = ["| Workload (BRDF 1, depth as configured) | mode | spp | schedule the tuner kept · kernel | **Msamples/s** | nodes / tris / hits per sample | fabric B/sample (read + write) | fabric rate · `frac` of 8 TB/s | L2 hit · requests/s = share of the measured 245 G/s · L1→L2 amplification | vector ALU busy × lanes = useful | waves waiting | `bound_measured` | algorithmic B/sample · `algorithmic_GBs` | CPU oracle (256 threads) | launch: events · rocprofv3 stats avg (calls) |",
        "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
import re
# <workload>_pN records (the runner-up schedule, pinned: counters for a bench line whose tuner chose it) are not rows of the table
order = sorted((k for k in d if not re.search(r"_p[0-6]$", k)), key=lambda k: (k.split("_walk")[0].replace("_native", ""), "_walk" in k, "_walk8c" in k, "_native" in k))
for key in order:
    r, b = d[key], d[key]["bench"]
    cfg, ps, roof = b["config"], b["per_sample"], b["roofline"]
    base = key.replace("_walk8c", "").replace("_walk8", "").replace("_walk6", "").replace("_native", "")
    mode = {"eight-order": "eight orders", "eight-order-compact": "eight orders, compact records", "six-order": "six orders"}.get(cfg["traversal"], "reference order") + (" + native" if cfg["arith"] == "native" else "")
    samples = cfg["width"] * cfg["height"] * b["steps"]
    p = r["pmc_timed_launch"]
    rd, wr = r.get("fabric_read_bytes_per_launch", 0.0), r.get("write_size_bytes", 0.0)
    ms = roof["launch_ms"]
    rate = (rd + wr) / (ms * 1e-3)
    req = p.get("TCC_REQ_sum", 0.0) / (ms * 1e-3)
    algo_launch = roof["algorithmic_bytes_per_launch"]
    issue = r.get("issue", {})
    fabric_frac = rate / 8e12
    l2_frac = req / L2_CEILING
    verdict = "fabric" if fabric_frac >= 0.5 else "l2-requests" if l2_frac >= 0.9 else "issue" if issue.get("valu_busy", 0) >= 0.7 else "latency"
    agree = r.get("timed_launch_agreement", {})
    cpu = b.get("cpu_baseline", {}).get("value")
    rows.append("| %s | %s | %d | %s · `%s` | **%.0f** | %.1f / %.1f / %.2f | %.0f + %.0f | %.2f TB/s = **%.2f** | %.0f %% · %.0f G/s = %.2f · %.1f | %.0f %% × %.0f %% = %.0f %% | %.0f %% | `%s` | %.0f · %.0f | %s | %.3f ms · %.3f ms (%d) |" % (
        LABEL.get(base, base), mode, b["steps"], b["schedule"], roof.get("kernel"), b["value"], ps["node_visits"], ps["triangle_tests"], ps["shaded_hits"],
        rd / samples, wr / samples, rate / 1e12, fabric_frac, 100 * r.get("l2_hit_rate", 0), req / 1e9, l2_frac, p.get("TCC_REQ_sum", 0.0) * 128.0 / algo_launch if algo_launch else 0.0,
        100 * issue.get("valu_busy", 0), 100 * issue.get("lane_utilisation", 0), 100 * issue.get("useful_lane_throughput_frac", 0),
        100 * r.get("wave_wait_fraction", 0), verdict, ps["algorithmic_bytes"], roof["algorithmic_GBs"],
        ("%.2f (%.0f×)" % (cpu, b["value"] / cpu)) if cpu else "—", ms, agree.get("rocprof_stats_average_ms", float("nan")), agree.get("rocprof_stats_calls", 0)))
text = "\n".join(rows)
print(text)
if "--write" in sys.argv:
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a, z = "<!-- %s-table-begin -->\n" % rnd, "<!-- %s-table-end -->" % rnd
    i, j = s.index(a) + len(a), s.index(z)
    open(path, "w").write(s[:i] + text + "\n" + s[j:])
