# pooled schedule: which waves have shading duty (all 12 = every wave), do they walk too, park share
for wt in 1 0; do for sh in 3 6 12; do for pk in 16 32; do
  echo "== walktoo $wt shaders $sh park $pk"; PBR_POOL_WALKTOO=$wt PBR_PH_PARK=$pk PBR_POOL_PATIENCE=16 PBR_POOL_SHADERS=$sh PBR_HIP_LIB=$PWD/lab/libpbrhip_pool.so timeout 100 python3 scripts/pool_check.py sponza:1920:1080:32 dragon:1920:1080:32 2>&1 | grep "pooled\|identical" | cut -c1-75
done; done; done
