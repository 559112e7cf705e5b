#!/bin/bash
# Collects the round's measurements on the GPU box: bench lines, rocprofv3 kernel stats and the PMC passes (each in its
# own run, each under a timeout; --pmc never together with a trace domain other than --kernel-trace).
# usage: scripts/profile_round.sh <outdir> [workload ...]     workloads: <scene>[_4k][_walk6|_walk8|_walk8c][_native][_pN], scene = cornell sponza dragon hairball
#        (_4k: 3840x2160; _walk6 / _walk8: pbr_config.traversal six / eight orders, _walk8c: eight orders over compact records; _native: pbr_config.arith native)
out=${1:-gpurun_out/round}; shift
loads=${@:-cornell sponza dragon hairball hairball_4k}
R=$PWD
mkdir -p $out
export TMPDIR=/tmp
for key in $loads; do
  s=${key%%_*}; size=""; steps=64
  [ $s = cornell ] && steps=256
  mode=""
  case $key in *_4k*) size="--width 3840 --height 2160"; steps=16;; esac
  case $key in *_walk6*) mode="$mode --traversal six-order";; *_walk8c*) mode="$mode --traversal eight-order-compact";; *_walk8*) mode="$mode --traversal eight-order";; esac
  case $key in *_native*) mode="$mode --arith native";; esac
  size="$size $mode"
  # <workload>_pN: the same workload with schedule N pinned — the runner-up where the tuner's call is close, so that a bench line
  # finds counters of whichever schedule it ran (bench.py, recorded_traffic)
  pin=""
  case $key in *_p[0-6]) pin="--plan ${key##*_p}";; esac
  timeout 600 python3 bench.py --scene $s $size --steps $steps --modes off $pin > $out/bench_$key.json 2> $out/bench_$key.err
  # the profiled runs pin the schedule the tuner settled on in the plain run (--plan: no tuning launches under the profiler)
  plan=$(python3 -c "import json,sys; n=json.loads(open('$out/bench_$key.json').read().strip().splitlines()[-1]).get('schedule','refill-lean'); print(['refill-lean','refill-wide','phased-lean','phased-wide','phased-mid','refill-mid','phased-dual'].index(n))")
  cd /tmp
  # warm-up as long as the timed render: every launch of the path-tracing kernel in kernel_stats.csv is then the same
  # work, and their average is comparable with the timed launch the bench line reports
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_$key -- python3 $R/bench.py --scene $s $size --steps $steps --warmup $steps --plan $plan --cpu-seconds 0 --hold-seconds 0 --modes off > $R/$out/stats_$key.json 2> /dev/null
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/pmc${i}_$key -- python3 $R/bench.py --scene $s $size --steps $steps --plan $plan --cpu-seconds 0 --hold-seconds 0 --modes off > /dev/null 2>&1 || echo "pmc pass $i failed for $key"
    i=$((i+1))
  done
  cd $R
done
python3 scripts/assemble_profiles.py --collect $out $loads
