cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "treelet or random_configurations" 2>&1 | tail -4
rm -f gpurun_out/r03f/layout.txt
for layout in 0 1 0 1; do
  PBR_NODE_LAYOUT=$layout PBR_PLAN=4 timeout 300 python scripts/ab.py sponza:64 dragon:64 hairball:32 2>&1 | sed "s/^/[layout $layout] /" >> gpurun_out/r03f/layout.txt
done
cat gpurun_out/r03f/layout.txt
