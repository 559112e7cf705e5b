cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03e/pytest_full.txt 2>&1
tail -5 gpurun_out/r03e/pytest_full.txt
rm -rf gpurun_out/round3
timeout 2400 bash scripts/profile_round.sh gpurun_out/round3 cornell sponza dragon hairball hairball_4k > gpurun_out/r03e/profile_round.log 2>&1
tail -6 gpurun_out/r03e/profile_round.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03e/bench_driver_command.json 2> gpurun_out/r03e/bench_driver_command.err
cat gpurun_out/r03e/bench_driver_command.json
timeout 600 python bench.py --gpus 2 --backend gloo --one-device --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r03e/bench_two_ranks_one_device.json 2> gpurun_out/r03e/bench_two_ranks.err
cat gpurun_out/r03e/bench_two_ranks_one_device.json; tail -3 gpurun_out/r03e/bench_two_ranks.err
python -c "import __graft_entry__ as g; g.smoke()"
