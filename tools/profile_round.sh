#!/bin/bash
# Measurements of record for a round, run on the GPU box from the repo root:  bash tools/profile_round.sh <outdir>
# (bench line; rocprofv3 kernel trace + stats of the headline and of the C5- / C4-shaped legs; PMC passes in runs of
# their own).  The summaries are copied to profiles/ by hand afterwards (profiles/README.md says which).
# SECTIONS="bench trace legs traffic valu" (the default: all of them) picks the parts to run.
set -o pipefail
OUT=${1:-gpurun_out/final}
SECTIONS=${SECTIONS:-bench trace legs traffic valu}
want() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}" || exit 1
B="python bench.py --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs"
if want bench; then
# the driver's literal command; the line on stdout, the full record in the detail file beside it
CHRONOCLUST_BENCH_DETAIL=$OUT/bench_detail.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
python - $OUT/bench.json <<'PY' || exit 1
import json, sys
text = open(sys.argv[1]).read()
lines = text.strip().splitlines()
assert len(lines) == 1, "stdout must carry exactly one line, got %d" % len(lines)
line = json.loads(lines[0])
assert len(lines[0]) < 4096, len(lines[0])
assert line["roofline"]["frac"] is not None and line["cpu_baseline"]["value"] > 0, "roofline / cpu_baseline missing"
print("bench line: %d bytes, value %.4g %s, %.2f ms/step, roofline.frac %.3f" % (len(lines[0]), line["value"], line["unit"], line["ms_per_step"], line["roofline"]["frac"]))
PY
echo "bench done"
fi
if want trace; then
CHRONOCLUST_BENCH_DETAIL=$OUT/bench_under_rocprof_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B --steps 4 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
python tools/prof_summary.py $OUT/trace > $OUT/kernel_summary.txt
python tools/scan_chain_summary.py $OUT/trace $OUT/bench_under_rocprof_detail.json > $OUT/scan_chain.json
echo "trace done"
CHRONOCLUST_BENCH_DETAIL=$OUT/bench_under_rocprof_nolookahead_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nola -o r -- $B --steps 4 --warmup 1 --lookahead 2 > $OUT/bench_under_rocprof_nolookahead.json 2> $OUT/trace_nola.err || exit 1
python tools/prof_summary.py $OUT/trace_nola > $OUT/kernel_summary_nolookahead.txt
echo "trace nola done"
fi
if want legs; then
# the two legs whose scans run at d = 40 / d = 14 (one_stream_exact: C5-shaped, events_sharded_relaxed: C4-shaped)
for LEG in one_stream_exact events_sharded_relaxed; do
  CHRONOCLUST_BENCH_DETAIL=$OUT/bench_leg_${LEG}_detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$LEG -o r -- python bench.py --only-leg $LEG --steps 1 --warmup 0 > $OUT/bench_leg_$LEG.json 2> $OUT/trace_$LEG.err || exit 1
  python tools/prof_summary.py $OUT/trace_$LEG > $OUT/kernel_summary_leg_$LEG.txt
  python tools/scan_chain_summary.py $OUT/trace_$LEG $OUT/bench_leg_${LEG}_detail.json $LEG > $OUT/scan_chain_leg_$LEG.json
done
echo "legs done"
fi
if want traffic; then
CHRONOCLUST_BENCH_DETAIL=$OUT/pmc_fetch_detail.json rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B --steps 1 --warmup 0 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
CHRONOCLUST_BENCH_DETAIL=$OUT/pmc_write_detail.json rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B --steps 1 --warmup 0 > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_detail.json $OUT/pmc_write_detail.json > $OUT/pmc_traffic.json
echo "pmc traffic done"
fi
if want valu; then
# VALU / LDS counters of the snapshot-scan kernels of full windows running alone (tools/steady.py, LA=2), at d = 20, 40,
# 14: the pruned chain (default policy) and the plain scan k_scan_u (CHRONOCLUST_HIP_PRUNE=0: what the start-up phase runs)
PA="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
PB="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
PC="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
# (VALU_SHAPES="20 5000 1000000" picks one shape: "<d> <microclusters> <points>;...")
IFS=';' read -ra SHAPES <<< "${VALU_SHAPES:-20 5000 1000000;40 50000 2000000;14 2000 2000000}"
for SH in "${SHAPES[@]}"; do
  set -- $SH
  for MODE in pruned plain; do
    if [ $MODE = plain ]; then export CHRONOCLUST_HIP_PRUNE=0; SUF=_plain; else unset CHRONOCLUST_HIP_PRUNE; SUF=; fi
    D=$1 G=$2 N=$3 LA=2 REPS=1 rocprofv3 --pmc $PA --output-format csv -d $OUT/pmc_valu_a_d$1$SUF -o r -- python tools/steady.py > $OUT/pmc_valu_a_d$1$SUF.txt 2>&1 || exit 1
    D=$1 G=$2 N=$3 LA=2 REPS=1 rocprofv3 --pmc $PB --output-format csv -d $OUT/pmc_valu_b_d$1$SUF -o r -- python tools/steady.py > $OUT/pmc_valu_b_d$1$SUF.txt 2>&1 || exit 1
    D=$1 G=$2 N=$3 LA=2 REPS=1 rocprofv3 --pmc $PC --output-format csv -d $OUT/pmc_valu_c_d$1$SUF -o r -- python tools/steady.py > $OUT/pmc_valu_c_d$1$SUF.txt 2>&1 || exit 1
    # (the window the policy settles on: the library's default, 49 152, or 32 768 while the table has fewer than a twelfth as many rows)
    W=49152; [ $(( $2 * 12 )) -lt 49152 ] && W=32768
    python tools/pmc_valu_summary.py $OUT/pmc_valu_a_d$1$SUF $OUT/pmc_valu_b_d$1$SUF $OUT/pmc_valu_c_d$1$SUF $1 $2 $W $OUT/pmc_valu_a_d$1$SUF.txt > $OUT/pmc_valu_d$1$SUF.json
  done
done
unset CHRONOCLUST_HIP_PRUNE
echo "pmc valu done"
fi
# (the raw traces and counter files are large: only the summaries made above travel back)
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls $OUT
