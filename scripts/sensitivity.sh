export PBR_PLAN=4
for s in 0 512 1024 2552; do echo "== LDS_SLOTS $s"; PBR_LDS_SLOTS=$s bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" base 2>&1 | grep Msamples; done
echo "== BLOCKS_PER_CU 1 (3 waves/SIMD)"; PBR_BLOCKS_PER_CU=1 bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16 cornell:64" base 2>&1 | grep Msamples
export PBR_PLAN=2
echo "== phased-lean (4 waves/SIMD, 1024 threads)"; bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" base 2>&1 | grep Msamples
export PBR_PLAN=3
echo "== phased-wide (8 waves/SIMD)"; bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" base 2>&1 | grep Msamples
