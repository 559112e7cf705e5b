set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "guard_build" 2>&1 | tail -5 > gpurun_out/r03a/pytest_guard.txt
cat gpurun_out/r03a/pytest_guard.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "drain_kernel or cornell_image or larger_scenes or every_tuner" 2>&1 | tail -15 > gpurun_out/r03a/pytest_subset.txt
cat gpurun_out/r03a/pytest_subset.txt
rm -f gpurun_out/r03a/ab.txt
for lib in lab/libpbrhip_r02.so ""; do
  PBR_HIP_LIB=$lib PBR_PLAN=4 timeout 300 python scripts/ab.py sponza:64 dragon:64 sponza:1 >> gpurun_out/r03a/ab.txt 2>&1
  PBR_HIP_LIB=$lib PBR_PLAN=5 timeout 300 python scripts/ab.py cornell:64 cornell:1 >> gpurun_out/r03a/ab.txt 2>&1
done
cat gpurun_out/r03a/ab.txt
timeout 900 python scripts/drain_ab.py sponza dragon > gpurun_out/r03a/drain_ab.txt 2>&1
cat gpurun_out/r03a/drain_ab.txt
