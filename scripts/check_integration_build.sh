#!/bin/bash
# Runs the build commands of INTEGRATION.md section 1 verbatim in a scratch directory and asks the result which modes it
# carries (pbr_mode_built): the full library all eight (traversal, arith) pairs, a link of pbr_hip.o + inst_f0_g*.o only (0, 0).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); cd $T
FLAGS="--offload-arch=gfx950:xnack- -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -I $R/include -I $R/physically-based-rendering_amd/csrc"
hipcc $FLAGS -c $R/physically-based-rendering_amd/csrc/pbr_hip.hip -o pbr_hip.o &
for f in 0 1 2 3 5 7; do for g in 0 1 2 3 4 5 6 7; do
  [ $g = 7 ] && [ $((f & 4)) != 0 ] && continue
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
    print(name, {(t, a): l.pbr_mode_built(t, a) for t in (0, 1, 2, 3) for a in (0, 1)})
PY
# the in-process multi-GPU driver and its example program, compiled and linked against RCCL
H=$R/physically-based-rendering_amd/host
g++ -O2 -std=c++17 -fPIC -shared -pthread -D__HIP_PLATFORM_AMD__ -I $R/include -I $H -I /opt/rocm/include $H/multi_path_tracer.cpp -o libpbrmulti.so \
    -L . -lpbrhip -L /opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,/opt/rocm/lib
g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -I $R/include -I $H -I /opt/rocm/include $H/Cfg.cpp $H/model_io.cpp $H/bvh_builder.cpp $H/scene_gen.cpp \
    $H/path_tracer.cpp $H/cl_adaptor.cpp $H/host_capi.cpp -o libpbrhost.so -L . -lpbrhip
g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I $R/include -I $H -I /opt/rocm/include $H/examples/multi_gpu_render.cpp -o multi_gpu_render \
    -L . -lpbrmulti -lpbrhost -lpbrhip -L /opt/rocm/lib -lrccl -lamdhip64 -pthread -Wl,-rpath,$T -Wl,-rpath,/opt/rocm/lib
echo "multi_gpu_render links against: $(readelf -d multi_gpu_render | grep -o 'lib[a-z0-9]*\.so[.0-9]*' | tr '\n' ' ')"
if [ "$1" = "--run" ]; then LD_LIBRARY_PATH=$T:$LD_LIBRARY_PATH ./multi_gpu_render cornell 16 640 360; fi
cd /; rm -rf $T
