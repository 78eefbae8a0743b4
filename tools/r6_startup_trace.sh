#!/bin/bash
# Round 6: the start-up stretch window by window (kernel trace) + the library's own batch lines.
set -o pipefail
OUT=${1:-gpurun_out/r6st}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
REPS=3 python3 tools/startup.py > $OUT/startup.txt 2>&1 || exit 1
REPS=2 CHRONOCLUST_HIP_TRACE=1 python3 tools/startup.py > $OUT/batches.txt 2>&1 || exit 1
REPS=2 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python3 tools/startup.py > $OUT/trace.txt 2>&1 || exit 1
python3 tools/gaps.py $OUT/trace > $OUT/gaps.txt
python3 tools/startup_timeline.py $OUT/trace 0 1000 ${DETAIL:-2,5,8,12,16,20,25,29,32,33,36} > $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/startup.txt; tail -30 $OUT/gaps.txt
