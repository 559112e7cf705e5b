cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cornell_image or larger_scenes or every_tuner or phong or guard_build or random_configurations or chunks or banded" 2>&1 | tail -4
timeout 900 python scripts/sweep_refill.py > gpurun_out/r03g/sweep_refill.txt 2>&1
cat gpurun_out/r03g/sweep_refill.txt
