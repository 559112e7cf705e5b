#!/bin/bash
# (the bench lines only: scripts/final_evidence.sh without the suites)  The round's closing evidence on the FINAL library, after its PMC passes are committed (profiles/rNN/pmc_traffic.json): every
# bench line then carries its own roofline.  usage: scripts/final_evidence.sh <outdir>
out=${1:-gpurun_out/final}; mkdir -p $out
loads="cornell cornell_native sponza sponza_walk8 sponza_walk8_native sponza_walk8c dragon dragon_walk8 dragon_walk8_native dragon_walk8c hairball hairball_4k hairball_4k_walk8 hairball_4k_walk8_native hairball_4k_walk8c"
for key in $loads; do
  s=${key%%_*}; size=""; steps=64
  [ $s = cornell ] && steps=256
  mode=""
  case $key in *_4k*) size="--width 3840 --height 2160"; steps=16;; esac
  case $key in *_walk8c*) mode="$mode --traversal eight-order-compact";; *_walk8*) mode="$mode --traversal eight-order";; esac
  case $key in *_native*) mode="$mode --arith native";; esac
  timeout 600 python3 bench.py --scene $s $size $mode --steps $steps --modes off --hold-seconds 0 > $out/line_$key.json 2> $out/line_$key.err
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command.json 2> $out/bench_driver_command.err
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --cpu-seconds 0 --hold-seconds 0 > $out/bench_force_dist.json 2> $out/bench_force_dist.err
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k bench > $out/pytest_bench.txt 2>&1
grep -E "passed|failed" $out/pytest_bench.txt
