#!/bin/bash
# usage: scripts/pmc.sh <outdir> <kernel-substring> -- python3 <script> [args]   (run from repo root on the GPU box)
# One rocprofv3 --pmc pass per counter set, each under its own timeout; prints per-kernel sums.
out=$1; kern=$2
# The command after `--` must be the PROGRAM ITSELF (python3 <script> ... or a binary): with --pmc the profiler's preload
# initialises the GPU before the program starts, so env / bash -c / taskset / numactl / a "#!/usr/bin/env" script would be
# an exec from a GPU-initialised process, which this pool refuses.  Set variables by exporting them before this script.
if [ "$3" != "--" ] || [ $# -lt 4 ]; then echo "usage: $0 <outdir> <kernel-substring> -- python3 <script> [args]" >&2; exit 2; fi
case "$(basename "$4")" in env|bash|sh|taskset|numactl|timeout|nice) echo "$0: '$4' re-execs: put the program itself after --" >&2; exit 2;; esac
if [ -f "$4" ] && head -c 64 "$4" | grep -q '^#!.*env'; then echo "$0: '$4' is a #!/usr/bin/env script: run it as python3 $4" >&2; exit 2; fi
shift 3
export TMPDIR=/tmp
mkdir -p $out
sets=(
"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum"
"TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum"
"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_CYCLE_sum"
"SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
)
i=0
for s in "${sets[@]}"; do
  timeout 150 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $out/set$i -- "$@" > $out/set$i.log 2>&1 || echo "set $i failed/timeout: $s"
  i=$((i+1))
done
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/set*/*/*_counter_collection.csv")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            k = (r["Dispatch_Id"], r["Counter_Name"])
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
            agg[(r["Dispatch_Id"], "ns")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    last = sorted({d for d, _ in agg}, key=int)[-1:]
    for (d, c), v in agg.items():
        if d in last:
            print("dispatch %s %-40s %.6g" % (d, c, v))
PY
