#!/bin/bash
# usage: scripts/pmc_one.sh <outdir> "<counters>" <kernel-substring> -- <command...>   one rocprofv3 --pmc pass
out=$1; ctrs=$2; kern=$3
# The command after `--` must be the PROGRAM ITSELF (see scripts/pmc.sh): no env / bash -c / taskset / #!/usr/bin/env hop.
if [ "$4" != "--" ] || [ $# -lt 5 ]; then echo "usage: $0 <outdir> \"<counters>\" <kernel-substring> -- python3 <script> [args]" >&2; exit 2; fi
case "$(basename "$5")" in env|bash|sh|taskset|numactl|timeout|nice) echo "$0: '$5' re-execs: put the program itself after --" >&2; exit 2;; esac
if [ -f "$5" ] && head -c 64 "$5" | grep -q '^#!.*env'; then echo "$0: '$5' is a #!/usr/bin/env script: run it as python3 $5" >&2; exit 2; fi
shift 4
export TMPDIR=/tmp
mkdir -p $out
timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/run -- "$@" > $out/run.log 2>&1 || echo "failed/timeout: $ctrs"
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/run/*/*_counter_collection.csv")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            k = (r["Dispatch_Id"], r["Counter_Name"])
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
            agg[(r["Dispatch_Id"], "ns")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for d in sorted({d for d, _ in agg}, key=int)[-4:]:
        print("dispatch", d, "  ".join("%s=%.5g" % (c, v) for (dd, c), v in agg.items() if dd == d))
PY
