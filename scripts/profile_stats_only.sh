#!/bin/bash
# Re-runs only the rocprofv3 --kernel-trace --stats pass of scripts/profile_round.sh with a given schedule pinned:
# scripts/profile_stats_only.sh <outdir> scene:plan ...      (plan = bench.py --plan index, 0..6)
# (this used to export PBR_PLAN, which bench.py no longer reads — the plan was silently not pinned.)
out=$1; shift
R=$PWD
mkdir -p $out
export TMPDIR=/tmp
for sp in "$@"; do
  s=${sp%%:*}; plan=${sp##*:}
  steps=64; [ $s = cornell ] && steps=256
  rm -rf $R/$out/stats_$s
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_$s -- python3 $R/bench.py --scene $s --steps $steps --warmup $steps --plan $plan --hold-seconds 0 --cpu-seconds 0 > $R/$out/stats_$s.json 2> /dev/null
  cd $R
  python3 -c "import json; b=json.loads(open('$out/stats_$s.json').read().strip().splitlines()[-1]); print('$s', b['schedule'], b['value'], b['roofline']['launch_ms'])"
done
