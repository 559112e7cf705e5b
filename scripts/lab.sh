#!/bin/bash
# Experiment builds: scripts/lab.sh name "-DPBR_EXP_..." [name2 "flags2" ...]  -> lab/libpbrhip_<name>.so
# The product's sources (all plans, all build flavours: physically-based-rendering_amd/build.py, build_lab) with the
# measurement hooks of lab/src/pt_lab_hooks.hpp compiled in (-DPBR_LAB_HOOKS) and the experiment's flags.  NOT the product build.
# (The node-phase and two-walk variants of round 4 — lab/src/pt_r04_*.hpp — were compiled into csrc/ under -DPBR_LAB up to
# round 4's last commit; round 5 took that plumbing out of the product sources: build them from `git worktree add ../r04 047ae5b`.)
set -e
cd "$(dirname "$0")/.."
mkdir -p lab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  python3 physically-based-rendering_amd/build.py --lab "$name" $flags
done
ls -la lab/
