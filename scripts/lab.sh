#!/bin/bash
# Experiment builds: scripts/lab.sh name "-DPBR_EXP_..." [name2 "flags2" ...]  -> lab/libpbrhip_<name>.so
# (PBR_LAB: only the kernel variants the bench scenes run; NOT the product build).
set -e
cd "$(dirname "$0")/.."
mkdir -p lab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -DPBR_LAB=1 -DPBR_LAB_HOOKS=1 $flags \
    -I include -I physically-based-rendering_amd/csrc -I lab/src -o lab/libpbrhip_$name.so physically-based-rendering_amd/csrc/pbr_hip.hip &
done
wait
ls -la lab/
