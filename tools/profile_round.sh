#!/bin/bash
# Measurements of record for a round, run on the GPU box from the repo root:  bash tools/profile_round.sh <outdir>
# (bench line, rocprofv3 kernel trace + stats of the same command, PMC passes in runs of their own).
set -o pipefail
OUT=${1:-gpurun_out/final}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}" || exit 1
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
B="python bench.py --no-cpu-baseline --no-one-stream --no-relaxed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B --steps 4 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
python tools/prof_summary.py $OUT/trace > $OUT/kernel_summary.txt
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nola -o r -- $B --steps 4 --warmup 1 --lookahead 2 > $OUT/bench_under_rocprof_nolookahead.json 2> $OUT/trace_nola.err || exit 1
python tools/prof_summary.py $OUT/trace_nola > $OUT/kernel_summary_nolookahead.txt
echo "trace nola done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B --steps 1 --warmup 0 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B --steps 1 --warmup 0 > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write 24576 > $OUT/pmc_traffic.json
echo "pmc traffic done"
# VALU counters of full-window clean scans running alone (tools/steady.py, LA=2), two passes of four counters
LA=2 REPS=1 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_valu_a -o r -- python tools/steady.py > $OUT/pmc_valu_a.txt 2>&1 || exit 1
LA=2 REPS=1 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_valu_b -o r -- python tools/steady.py > $OUT/pmc_valu_b.txt 2>&1 || exit 1
python tools/pmc_valu_summary.py $OUT/pmc_valu_a $OUT/pmc_valu_b > $OUT/pmc_valu.json
echo "pmc valu done"
ls $OUT
