# the published empty-heads mask (lab/libpbrhip_published.so) against the product: correctness, then single frames, shares, long launches
PBR_LAB_ENV=1 PBR_HIP_LIB=lab/libpbrhip_published.so timeout 600 python -m pytest tests/test_gpu_deal_order.py tests/test_gpu_parity.py -x -q -m gpu -k "not bench" 2>&1 | grep -E "passed|failed"
for rep in 1 2; do
for lib in physically-based-rendering_amd/csrc/libpbrhip.so lab/libpbrhip_published.so; do
  echo "== $lib"
  PBR_LAB_ENV=1 PBR_HIP_LIB=$lib python scripts/frame_latency.py 2>&1 | grep "single-frame"
  for s in sponza dragon; do PBR_LAB_ENV=1 PBR_HIP_LIB=$lib timeout 300 python scripts/shard_scaling.py $s 20 | grep "N=[18]"; done
done
done
HEADS_LIBS="physically-based-rendering_amd/csrc/libpbrhip.so lab/libpbrhip_published.so" bash scripts/heads_ab.sh
