#!/bin/bash
# Round 6: k_scan_p3 with the prefix test over up to 24 dimensions (second MFMA): C2 and the C5 shape, steady state
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for la in 2 0; do
echo "== C2 LA=$la"; LA=$la REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [2]" -A1
echo "== C5 shape LA=$la"; D=40 G=50000 N=2000000 LA=$la REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]" -A1
echo "== C5 shape, never listed LA=$la"; CHRONOCLUST_HIP_P3_LISTED=1000000000 D=40 G=50000 N=2000000 LA=$la REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]"
done
timeout -k 10 600 python3 -m pytest tests/test_pruned_scan.py -x -q -m gpu 2>&1 | tail -3
