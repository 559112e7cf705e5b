#!/bin/bash
# Runs the build commands of INTEGRATION.md section 1 verbatim in a scratch directory and asks the result which modes it
# carries (pbr_mode_built): the full library all six (traversal, arith) pairs, a link of pbr_hip.o + inst_f0_g*.o only (0, 0).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); cd $T
FLAGS="--offload-arch=gfx950:xnack- -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -I $R/include -I $R/physically-based-rendering_amd/csrc"
hipcc $FLAGS -c $R/physically-based-rendering_amd/csrc/pbr_hip.hip -o pbr_hip.o &
for f in 0 1 2 3; do for g in 0 1 2 3 4 5 6 7; do
  [ $g = 3 ] && [ $f != 0 ] && continue
  NATIVE=""; [ $((f & 2)) != 0 ] && NATIVE="-fno-hip-fp32-correctly-rounded-divide-sqrt"
  hipcc $FLAGS $NATIVE -DPT_FLAVOUR=$f -DPT_GROUP=$g -c $R/physically-based-rendering_amd/csrc/pt_instance.hip -o inst_f${f}_g${g}.o &
done; done; wait
hipcc --offload-arch=gfx950:xnack- -shared -fPIC pbr_hip.o inst_f*_g*.o -o libpbrhip.so
hipcc --offload-arch=gfx950:xnack- -shared -fPIC pbr_hip.o inst_f0_g*.o -o libpbrhip_ref.so
python3 - <<PY
import ctypes
for name in ("libpbrhip.so", "libpbrhip_ref.so"):
    l = ctypes.CDLL("$T/" + name)
    l.pbr_mode_built.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    print(name, {(t, a): l.pbr_mode_built(t, a) for t in (0, 1, 2) for a in (0, 1)})
PY
cd /; rm -rf $T
