cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
R=$PWD
cd /tmp
for s in sponza hairball; do
bash $R/scripts/pmc.sh $R/gpurun_out/r03d/pmc_tcp_$s pathTracing -- python3 $R/bench.py --scene $s --steps 32 --plan 4 --cpu-seconds 0 > $R/gpurun_out/r03d/pmc_tcp_$s.txt 2>&1
done
cat $R/gpurun_out/r03d/pmc_tcp_sponza.txt $R/gpurun_out/r03d/pmc_tcp_hairball.txt
