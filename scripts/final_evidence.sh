#!/bin/bash
# The round's closing evidence on the FINAL library, after its PMC passes are committed (profiles/rNN/pmc_traffic.json): every
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
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
bash scripts/full_configs.sh > $out/full_configs.txt 2>&1
for s in sponza dragon hairball; do timeout 300 python3 scripts/shard_scaling.py $s 20; AB_TRAVERSAL=2 timeout 300 python3 scripts/shard_scaling.py $s 20; done > $out/shard_scaling_steps20.txt 2>&1
python3 scripts/frame_latency.py > $out/frame_latency.txt 2>&1
PBR_SOAK_SEEDS=6000 timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k random_configurations_bit_exact > $out/soak.txt 2>&1
PBR_WALK_SOAK_SEEDS=9000 timeout 900 python3 -m pytest tests/test_gpu_walk_order.py -q -m gpu -k random_configurations_in_an_ordered_mode > $out/soak_walk.txt 2>&1
PBR_NATIVE_SOAK_SEEDS=2000 timeout 600 python3 -m pytest tests/test_gpu_native_arith.py -q -m gpu -k random_configurations_in_the_native > $out/soak_native.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1
for f in pytest_gpu soak soak_walk soak_native smoke; do tail -n 3 $out/$f.txt; done
