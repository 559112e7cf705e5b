cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cornell_image or larger_scenes or every_tuner or reference_scenes or lights" 2>&1 | tail -4 > gpurun_out/r03b/pytest_subset.txt
cat gpurun_out/r03b/pytest_subset.txt
rm -f gpurun_out/r03b/ab.txt
for lib in lab/libpbrhip_r02.so "" lab/libpbrhip_r02.so ""; do
  PBR_HIP_LIB=$lib PBR_PLAN=4 timeout 300 python scripts/ab.py sponza:64 dragon:64 sponza:1 >> gpurun_out/r03b/ab.txt 2>&1
  PBR_HIP_LIB=$lib PBR_PLAN=2 timeout 300 python scripts/ab.py hairball:32 >> gpurun_out/r03b/ab.txt 2>&1
  PBR_HIP_LIB=$lib PBR_PLAN=5 timeout 300 python scripts/ab.py cornell:64 >> gpurun_out/r03b/ab.txt 2>&1
  PBR_HIP_LIB=$lib PBR_PLAN=3 timeout 300 python scripts/ab.py sponza:64 >> gpurun_out/r03b/ab.txt 2>&1
done
cat gpurun_out/r03b/ab.txt
