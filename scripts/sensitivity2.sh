# what does scratch cost at equal occupancy?  mid kernel (80 VGPRs, 64 B scratch / lane) as 512-thread blocks, 2 per CU
# (PBR_BLOCKS_PER_CU=2) = 4 waves / SIMD, against the lean kernel (104 VGPRs, no scratch) at 4 waves / SIMD
export PBR_PLAN=4
echo "== phased-mid kernel, 512-thread blocks x 2 = 4 waves/SIMD"; PBR_BLOCKS_PER_CU=2 bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" mid512 2>&1 | grep Msamples
echo "== phased-mid kernel, 512-thread blocks x 3 = 6 waves/SIMD"; bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" mid512 2>&1 | grep Msamples
echo "== phased-mid kernel, 256-thread blocks x 6 = 6 waves/SIMD"; bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" mid256 2>&1 | grep Msamples
export PBR_PLAN=2
echo "== phased-lean kernel 4 waves/SIMD"; bash scripts/lab_run.sh "sponza:32 dragon:32 hairball:16" base 2>&1 | grep Msamples
