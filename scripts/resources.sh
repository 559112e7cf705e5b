#!/bin/bash
# Register / scratch / LDS use of every path-tracing kernel for a set of -D flags: scripts/resources.sh "-DPBR_LAB=1 -DFOO"
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -c $1 \
  -Rpass-analysis=kernel-resource-usage -I include -I physically-based-rendering_amd/csrc -o /dev/null physically-based-rendering_amd/csrc/pbr_hip.hip 2>&1 |
python3 -c '
import re, sys
name = None; row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); row = {}
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]", "SGPRs"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m and name:
            row[key.split(" ")[0]] = int(m.group(1))
    if name and "LDS" in row:
        if "pathTracing" in name:
            print("%-72s VGPR %3d  SGPR %3d  scratch %4d  occupancy %d  LDS %d" % (name[:72], row.get("VGPRs", -1), row.get("SGPRs", -1), row.get("ScratchSize", -1), row.get("Occupancy", -1), row["LDS"]))
        name = None
'
