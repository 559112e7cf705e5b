# lab: does the number of queue heads matter at the END of a launch (all XCDs on the last bands' heads)?  product (8 heads) vs a lab
# build with 16 (two per XCD), spatial vs expensive-last dealing, plan pinned
for lib in ${HEADS_LIBS:-physically-based-rendering_amd/csrc/libpbrhip.so lab/libpbrhip_heads16.so}; do
  for sc in "cornell 5 256" "sponza 6 64" "dragon 4 64" "hairball 4 32"; do
    set -- $sc
    for order in 0 2; do
      PBR_LAB_ENV=1 PBR_HIP_LIB=$lib PBR_DEAL_ORDER=$order python bench.py --scene $1 --plan $2 --steps $3 --cpu-seconds 0 --hold-seconds 0 --modes off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %-8s deal %-14s %8.1f Msamples/s  %s' % ('$lib', d['config']['scene'], d['deal'], d['value'], d['schedule']))"
    done
  done
done
# the driver's command (the library picks plan and order), three times per library
for lib in ${HEADS_LIBS:-physically-based-rendering_amd/csrc/libpbrhip.so lab/libpbrhip_heads16.so}; do
  for rep in 1 2 3; do
    PBR_LAB_ENV=1 PBR_HIP_LIB=$lib python bench.py --cpu-seconds 0 --hold-seconds 0 --modes off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %-8s deal %-14s %8.1f Msamples/s  %s  (driver command)' % ('$lib', d['config']['scene'], d['deal'], d['value'], d['schedule']))"
  done
done
