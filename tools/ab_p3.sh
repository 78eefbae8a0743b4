#!/bin/bash
# Round 6: k_scan_p3 at the C5 shape (2 M x 40, 50 000 microclusters) and at C2: kept rows listed or completed at once; the round-5 forms
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for la in 2 0; do
echo "== C2 LA=$la"; LA=$la REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [2]"
echo "== C5 shape, LISTED never LA=$la"; CHRONOCLUST_HIP_P3_LISTED=1000000000 D=40 G=50000 N=2000000 LA=$la REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]"
echo "== C5 shape, LISTED always LA=$la"; CHRONOCLUST_HIP_P3_LISTED=0 D=40 G=50000 N=2000000 LA=$la REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]"
echo "== C5 shape, round-5 kernels (k_scan_a + k_scan_p<MASKED>) LA=$la"; CHRONOCLUST_HIP_SCANP3=0 D=40 G=50000 N=2000000 LA=$la REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]"
done
