cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "guard_build" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "idle_lanes or cooperative or cornell_image or larger_scenes" 2>&1 | tail -8
timeout 900 python scripts/end_of_launch.py sponza dragon --coop 0,8,16 --eighths 4,8 > gpurun_out/r03c/end_of_launch.txt 2>&1
cat gpurun_out/r03c/end_of_launch.txt
