#!/bin/bash
# Round 6, records of the final tree (second part): the snapshot-scan kernels side by side at the C2 / C5 / C4 shapes, the other
# workloads of README.md (C3, skewed streams), the C5-shaped full-length oracle check.
set -o pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
O=gpurun_out/r06_tool_scan_p3.txt
: > $O
run() { echo "== $1" >> $O; shift; env "$@" python3 tools/steady.py 2>&1 | grep "steady run [1]" -A1 >> $O; }
for la in 2 0; do
run "C2 (1 M x 20, 5 000 rows), k_scan_p3 (default), LA=$la" LA=$la REPS=2
run "C2, k_scan_p2 (prefix test on the VALU, two points per lane), LA=$la" LA=$la REPS=2 CHRONOCLUST_HIP_SCANP3=0 CHRONOCLUST_HIP_LA_PRUNED=1
run "C2, k_scan_p (one point per lane: round 5), LA=$la" LA=$la REPS=2 CHRONOCLUST_HIP_SCANP3=0 CHRONOCLUST_HIP_SCANP2=0 CHRONOCLUST_HIP_LA_PRUNED=1
run "C2, k_scan_p3 with the kept rows listed, LA=$la" LA=$la REPS=2 CHRONOCLUST_HIP_P3_LISTED=0
run "C5 shape (2 M x 40, 50 000 rows), k_scan_p3 (default: kept rows listed), LA=$la" LA=$la REPS=2 D=40 G=50000 N=2000000
run "C5 shape, k_scan_p3, kept rows completed at once, LA=$la" LA=$la REPS=2 D=40 G=50000 N=2000000 CHRONOCLUST_HIP_P3_LISTED=1000000000
run "C5 shape, k_scan_a + k_scan_p<MASKED> (round 5), LA=$la" LA=$la REPS=2 D=40 G=50000 N=2000000 CHRONOCLUST_HIP_SCANP3=0 CHRONOCLUST_HIP_LA_PRUNED=1
run "C4 shape (2 M x 14, 2 000 rows), k_scan_p3, LA=$la" LA=$la REPS=2 D=14 G=2000 N=2000000
run "C4 shape, round-5 kernels, LA=$la" LA=$la REPS=2 D=14 G=2000 N=2000000 CHRONOCLUST_HIP_SCANP3=0 CHRONOCLUST_HIP_LA_PRUNED=1
done
cat $O
S=gpurun_out/r06_tool_shapes.txt
( echo "== tools/c3.py (C3: 5 x 1 M x 20, drift + churn, both trackers)"; python3 tools/c3.py 2>&1 | tail -12
  echo "== tools/skewed.py (2 M x 14, 2 000 populations, three take 30 % of the events)"; N=2000000 D=14 G=2000 HEAVY=0.3 python3 tools/skewed.py 2>&1 | tail -4 ) > $S 2>&1
cat $S | cut -c1-220
python3 tools/full_oracle.py c5tail > gpurun_out/r06final_oracle_c5tail.txt 2>&1; echo "full_oracle c5tail rc $?"; tail -2 gpurun_out/r06final_oracle_c5tail.txt
