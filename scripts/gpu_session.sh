cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03j/pytest_full.txt 2>&1
tail -4 gpurun_out/r03j/pytest_full.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03j/bench_driver_command.json 2> gpurun_out/r03j/bench.err
cut -c1-260 gpurun_out/r03j/bench_driver_command.json
python -c "import __graft_entry__ as g; g.smoke()"
