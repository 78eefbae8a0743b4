#!/bin/bash
# sweep of the sub-range split of pruned scans (CHRONOCLUST_HIP_PRUNE_WGS = workgroups per CU) on the steady state
export CHRONOCLUST_HIP_PRUNE=2
for W in ${WGS:-2 4 8 12 16}; do
  export CHRONOCLUST_HIP_PRUNE_WGS=$W
  echo "== WGS=$W alone"; LA=2 REPS=1 python tools/steady.py 2>&1 | grep -A1 "steady run"
  echo "== WGS=$W lookahead"; LA=0 REPS=1 python tools/steady.py 2>&1 | grep "steady run"
done
