#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for v in 8 16 32 64; do
  echo "=== windows per sync $v"
  WPS=$v N=4000000 LA=0 REPS=2 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
done
