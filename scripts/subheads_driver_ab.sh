# one head per band (lab/libpbrhip_sub1.so = the sources of commit 8583261 with -DPT_SUB=1) vs four (the product), same box, alternating:
# the driver's command, then the 8-way share at --steps 20 (scripts/shard_scaling.py)
for rep in 1 2 3 4 5; do
  for lib in lab/libpbrhip_sub1.so physically-based-rendering_amd/csrc/libpbrhip.so; do
    PBR_LAB_ENV=1 PBR_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --hold-seconds 0 --modes off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %-8s deal %-14s %8.1f Msamples/s  %s' % ('$lib', d['config']['scene'], d['deal'], d['value'], d['schedule']))"
  done
done
for lib in lab/libpbrhip_sub1.so physically-based-rendering_amd/csrc/libpbrhip.so; do
  echo "== $lib"
  for s in sponza dragon; do PBR_LAB_ENV=1 PBR_HIP_LIB=$lib timeout 300 python scripts/shard_scaling.py $s 20; done
done
