#!/bin/bash
# Validation kernels and lookahead scans on disjoint sets of CUs (CHRONOCLUST_HIP_VAL_CUS): steady state.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for V in ${VALS:-0 64 96 128}; do
  echo "== CHRONOCLUST_HIP_VAL_CUS=$V"
  CHRONOCLUST_HIP_VAL_CUS=$V LA=0 REPS=3 timeout -k 10 120 python tools/steady.py 2>&1 | grep "steady run" | tail -1 | cut -c1-150 || { echo "timed out / failed"; exit 1; }
done
