#!/bin/bash
# Round 6: the numbers the round starts from (bench line, start-up stretch with its kernel trace and gaps, steady state).
set -o pipefail
OUT=${1:-gpurun_out/r6base}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
cp bench_detail.json $OUT/bench_detail.json
REPS=3 python3 tools/startup.py > $OUT/startup.txt 2>&1 || exit 1
REPS=2 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python3 tools/startup.py > $OUT/trace.txt 2>&1 || exit 1
python3 tools/gaps.py $OUT/trace > $OUT/gaps.txt
python3 tools/kernel_avgs.py $OUT/trace > $OUT/kernel_avgs.txt
REPS=3 python3 tools/steady.py > $OUT/steady.txt 2>&1 || exit 1
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/bench.json | cut -c1-600; cat $OUT/startup.txt $OUT/gaps.txt; tail -4 $OUT/steady.txt
