cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 300 scripts/micro/valu_rate > gpurun_out/r03c/valu_rate.txt 2>&1
grep -E "v_fma|v_add_f32 |MI3" gpurun_out/r03c/valu_rate.txt
