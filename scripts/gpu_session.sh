cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
( time PBR_EXHAUSTIVE=1 timeout 1500 python -m pytest tests/test_gpu_math_exhaustive.py -x -q -m gpu ) > gpurun_out/r03k/math_exhaustive.txt 2>&1
tail -6 gpurun_out/r03k/math_exhaustive.txt
( time PBR_SOAK_SEEDS=6000 timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k random_configurations ) > gpurun_out/r03k/soak.txt 2>&1
tail -6 gpurun_out/r03k/soak.txt
