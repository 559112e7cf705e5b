cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
rm -f gpurun_out/r03i/policy.txt
for lib in "" lab/libpbrhip_nt.so lab/libpbrhip_sc0.so lab/libpbrhip_sc1.so lab/libpbrhip_sc01.so ""; do
  PBR_HIP_LIB=$lib PBR_PLAN=4 timeout 300 python scripts/ab.py sponza:32 dragon:32 hairball:16 >> gpurun_out/r03i/policy.txt 2>&1
done
cat gpurun_out/r03i/policy.txt
