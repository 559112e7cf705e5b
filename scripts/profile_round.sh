#!/bin/bash
# Collects the round's measurements on the GPU box: bench lines, rocprofv3 kernel stats and the
# memory PMC passes (each in its own run, each under a timeout).  usage: scripts/profile_round.sh <outdir>
out=${1:-gpurun_out/round}
R=$PWD
mkdir -p $out
export TMPDIR=/tmp
for s in cornell sponza dragon hairball; do
  steps=64; [ $s = cornell ] && steps=256
  timeout 300 python3 bench.py --scene $s --steps $steps > $out/bench_$s.json 2> $out/bench_$s.err
  # the profiled runs use the schedule the tuner settled on in the plain run (PBR_PLAN: no tuning launches under the profiler)
  plan=$(python3 -c "import json,sys; n=json.loads(open('$out/bench_$s.json').read().strip().splitlines()[-1]).get('schedule','refill-lean'); print(['refill-lean','refill-wide','phased-lean','phased-wide','phased-mid','refill-mid'].index(n))")
  export PBR_PLAN=$plan
  cd /tmp
  # warm-up as long as the timed render: the two launches of the path-tracing kernel in kernel_stats.csv are then the
  # same work, and their average is comparable with the timed launch the bench line reports
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_$s -- python3 $R/bench.py --scene $s --steps $steps --warmup $steps --cpu-seconds 0 > $R/$out/stats_$s.json 2> /dev/null
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc${i}_$s -- python3 $R/bench.py --scene $s --steps $steps --cpu-seconds 0 > /dev/null 2>&1 || echo "pmc pass $i failed for $s"
    i=$((i+1))
  done
  cd $R
  unset PBR_PLAN
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summary = {}
for s in ("cornell", "sponza", "dragon", "hairball"):
    rec = {}
    try:
        rec["bench"] = json.loads(open("%s/bench_%s.json" % (out, s)).read().strip().splitlines()[-1])
    except Exception as e:
        rec["bench_error"] = str(e)
    try:
        rec["bench_under_rocprof"] = json.loads(open("%s/stats_%s.json" % (out, s)).read().strip().splitlines()[-1])
        rows = list(csv.DictReader(open(glob.glob("%s/stats_%s/*/*_kernel_stats.csv" % (out, s))[0])))
        rec["kernel_stats"] = [r for r in rows if "ptk::" in r["Name"]]
    except Exception as e:
        rec["stats_error"] = str(e)
    pmc = collections.OrderedDict()
    for f in sorted(glob.glob("%s/pmc*_%s/*/*_counter_collection.csv" % (out, s))):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "pathTracing" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                per[int(r["Dispatch_Id"])]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if per:
            timed = per[max(per)]          # the last pathTracing dispatch is the timed launch
            for k, v in timed.items():
                pmc[k if k != "_ns" else "duration_ns(" + "+".join(c for c in timed if c != "_ns")[:40] + ")"] = v
    rec["pmc_timed_launch"] = pmc
    if "TCC_EA0_RDREQ_128B_sum" in pmc:
        rd = 128 * pmc["TCC_EA0_RDREQ_128B_sum"] + 64 * pmc.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * pmc.get("TCC_EA0_RDREQ_32B_sum", 0)
        rec["fabric_read_bytes_per_launch"] = rd
        rec["fetch_size_x2_bytes"] = 2 * 1024 * pmc.get("FETCH_SIZE", 0)
        rec["write_size_bytes"] = 1024 * pmc.get("WRITE_SIZE", 0)
        rec["l2_hit_rate"] = pmc.get("TCC_HIT_sum", 0) / max(1.0, pmc.get("TCC_HIT_sum", 0) + pmc.get("TCC_MISS_sum", 0))
    if "SQ_ACTIVE_INST_VALU" in pmc:
        rec["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (64 * pmc["SQ_ACTIVE_INST_VALU"])
        rec["wave_wait_fraction"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
    summary[s] = rec
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
for s, rec in summary.items():
    b = rec.get("bench", {})
    print(s, "Msamples/s %.1f" % b.get("value", 0), "roofline %.0f GB/s (%.0f %%)" % (b.get("roofline", {}).get("achieved", 0), 100 * b.get("roofline", {}).get("frac", 0)),
          "cpu %.2f" % b.get("cpu_baseline", {}).get("value", 0),
          "fabric read %.1f GB/launch" % (rec.get("fabric_read_bytes_per_launch", 0) / 1e9), "L2 hit %.3f" % rec.get("l2_hit_rate", 0),
          "lane util %.3f wait %.3f" % (rec.get("valu_lane_utilisation", 0), rec.get("wave_wait_fraction", 0)))
PY
