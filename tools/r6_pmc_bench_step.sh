#!/bin/bash
# The roofline counted on the timed run: VALU instructions of one bench step, per kernel (tools/pmc_bench_step.py).
#   bash tools/r6_pmc_bench_step.sh <outdir>      (GPU box; the program itself follows `--`: no wrapper)
set -o pipefail
OUT=${1:-gpurun_out/pmc_bench_step}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
CHRONOCLUST_BENCH_DETAIL=$OUT/detail.json rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o r -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 tools/pmc_bench_step.py $OUT/pmc $OUT/detail.json 2 > $OUT/pmc_bench_step.json || exit 1
find $OUT -name "*counter_collection.csv" -delete
python3 - $OUT/pmc_bench_step.json <<'PY'
import json, sys
p = json.load(open(sys.argv[1]))
print("one step: %.3g instruction-lanes over all kernels, %.3g in the snapshot scans (%d launches); by kernel:" % (
    p["all_kernels"]["instruction_lanes_per_step"], p["snapshot_scan"]["instruction_lanes_per_step"], p["snapshot_scan"]["launches_per_step"]))
for k, v in list(p["kernels"].items())[:14]:
    print("  %-44s %7.1f launches %10.3g lanes %8.1f us busy %s" % (k[:44], v["launches_per_step"], v["instruction_lanes_per_step"], v["us_per_step_under_pmc"], v["valu_busy_fraction"]))
PY
