#!/bin/bash
# usage: scripts/sweep.sh "jobs" sched:variant ...   (product library)
cd "$(dirname "$0")/.."
jobs=$1; shift
for sv in "$@"; do
  IFS=: read s v <<< "$sv"
  PBR_SCHEDULE=$s PBR_VARIANT=$v timeout 300 python3 scripts/ab.py $jobs 2>&1 | sed "s/^/[$v] /" | tail -8
done
