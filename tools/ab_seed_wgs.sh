#!/bin/bash
# k_seed2 at 4 / 5 / 6 / 8 workgroups per CU (build variants, CHRONOCLUST_HIP_SEED=2: sub-ranges sized to fill the machine once)
for W in 4 5 6 8; do
  L=chronoclust_amd/libcc_seedw$W.so; [ $W = 4 ] && L=chronoclust_amd/libchronoclust_hip.so
  echo "== CC_SEED2_WGS=$W"
  CHRONOCLUST_HIP_LIB=$PWD/$L CHRONOCLUST_HIP_SEED=2 LA=2 REPS=1 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_LIB=$PWD/$L CHRONOCLUST_HIP_SEED=2 LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_LIB=$PWD/$L CHRONOCLUST_HIP_SEED=2 N=2000000 D=40 G=50000 LA=2 REPS=1 python tools/steady.py 2>&1 | grep "steady run"
done
