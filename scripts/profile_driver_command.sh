# rocprofv3 kernel stats of the DRIVER's command (bench.py --gpus 1 --steps 20 --warmup 5), schedule pinned to the one its tuner keeps,
# so that the launch duration in the driver's line has its own profiler record: -> <outdir>/stats_driver/, <outdir>/stats_driver.json
out=${1:-gpurun_out/driver}; mkdir -p $out
R=$PWD
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --hold-seconds 0 --modes off > $out/plain.json 2> $out/plain.err
plan=$(python3 -c "import json; n=json.loads(open('$out/plain.json').read().strip().splitlines()[-1])['schedule']; print(['refill-lean','refill-wide','phased-lean','phased-wide','phased-mid','refill-mid','phased-dual'].index(n))")
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 20 --plan $plan --cpu-seconds 0 --hold-seconds 0 --modes off > $R/$out/stats_driver.json 2> /dev/null
cd $R
python3 - <<PY
import csv, glob, json
line = json.loads(open("$out/stats_driver.json").read().strip().splitlines()[-1])
rows = [r for r in csv.DictReader(open(glob.glob("$out/stats_driver/*/*_kernel_stats.csv")[0])) if "pathTracing" in r["Name"]]
print("bench line under rocprofv3: %.1f Msamples/s, %s, launch %.3f ms (HIP events)" % (line["value"], line["schedule"], line["roofline"]["launch_ms"]))
for r in rows:
    print("kernel_stats: %s calls %s avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
