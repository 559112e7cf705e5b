cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py -x -q -m gpu 2>&1 | tail -4
rm -f gpurun_out/r03d/ab.txt
for lib in lab/libpbrhip_r02.so "" lab/libpbrhip_r02.so ""; do
  PBR_HIP_LIB=$lib PBR_PLAN=4 timeout 300 python scripts/ab.py sponza:64 dragon:64 >> gpurun_out/r03d/ab.txt 2>&1
  PBR_HIP_LIB=$lib PBR_PLAN=2 timeout 300 python scripts/ab.py hairball:32 >> gpurun_out/r03d/ab.txt 2>&1
  PBR_HIP_LIB=$lib PBR_PLAN=5 timeout 300 python scripts/ab.py cornell:64 >> gpurun_out/r03d/ab.txt 2>&1
done
cat gpurun_out/r03d/ab.txt
