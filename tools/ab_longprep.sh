#!/bin/bash
# A/B of the laid-out long chains (CHRONOCLUST_HIP_LONGPREP, round 5) on the streams that have long chains: few microclusters
# (every chain long, one workgroup per table row) and skewed populations on a large table (listed chains).
for P in 0 1; do
  echo "== CHRONOCLUST_HIP_LONGPREP=$P"
  export CHRONOCLUST_HIP_LONGPREP=$P
  for cfg in "20 12" "20 50" "20 200" "20 1000" "14 30" "14 100" "5 12"; do
    set -- $cfg
    N=500000 D=$1 G=$2 REPS=2 timeout -k 10 120 python tools/one_regime.py | tail -1 || exit 1
  done
  N=2000000 timeout -k 10 120 python tools/skewed.py | tail -1 || exit 1
  N=2000000 HEAVY=0.05 timeout -k 10 120 python tools/skewed.py | tail -1 || exit 1
done
