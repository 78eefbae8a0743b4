#!/bin/bash
# Round 6, records of the final tree: start-up stretch (timeline, gaps), steady state, a bench step window by window, the
# full-length oracle checks, a soak of random cases.
set -o pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
export TMPDIR=/tmp
bash tools/r6_baseline.sh gpurun_out/r06final_base > gpurun_out/r06final_base.log 2>&1 || exit 1
bash tools/r6_startup_trace.sh gpurun_out/r06final_st > gpurun_out/r06final_st.log 2>&1 || exit 1
bash tools/r6_bench_trace.sh gpurun_out/r06final_bt > gpurun_out/r06final_bt.log 2>&1 || exit 1
for t in c2 c5tail c5plain; do python3 tools/full_oracle.py $t > gpurun_out/r06final_oracle_$t.txt 2>&1; echo "full_oracle $t rc $?"; tail -2 gpurun_out/r06final_oracle_$t.txt; done
