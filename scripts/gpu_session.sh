cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03h/pytest_full.txt 2>&1
tail -5 gpurun_out/r03h/pytest_full.txt
rm -rf gpurun_out/round3c
timeout 1200 bash scripts/profile_round.sh gpurun_out/round3c cornell > gpurun_out/r03h/profile_cornell.log 2>&1
tail -3 gpurun_out/r03h/profile_cornell.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03h/bench_driver_command.json 2> gpurun_out/r03h/bench_driver_command.err
cut -c1-300 gpurun_out/r03h/bench_driver_command.json
timeout 600 python bench.py --scene cornell --steps 256 --cpu-seconds 0 > gpurun_out/r03h/bench_cornell.json 2>/dev/null
cut -c1-200 gpurun_out/r03h/bench_cornell.json
