#!/bin/bash
# Runs scripts/ab.py for every lab/libpbrhip_*.so given by name: scripts/lab_run.sh "jobs" name1 name2 ...
cd "$(dirname "$0")/.."
jobs=$1; shift
for n in "$@"; do
  PBR_HIP_LIB=$PWD/lab/libpbrhip_$n.so timeout 300 python3 scripts/ab.py $jobs 2>&1 | tail -8
done
