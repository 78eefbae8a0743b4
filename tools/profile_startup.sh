#!/bin/bash
# The validation kernels over the start-up stretch of C2 (tools/startup.py): kernel trace (gaps) + three PMC passes.
#   bash tools/profile_startup.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/startup}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
N0=${N0:-200000}
REPS=3 N0=$N0 python tools/startup.py > $OUT/startup.txt 2>&1 || exit 1
REPS=2 N0=$N0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python tools/startup.py > $OUT/trace.txt 2>&1 || exit 1
python tools/gaps.py $OUT/trace > $OUT/gaps.txt
PA="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
PB="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
PC="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
REPS=2 N0=$N0 rocprofv3 --pmc $PA --output-format csv -d $OUT/pmc_a -o r -- python tools/startup.py > $OUT/pmc_a.txt 2>&1 || exit 1
REPS=2 N0=$N0 rocprofv3 --pmc $PB --output-format csv -d $OUT/pmc_b -o r -- python tools/startup.py > $OUT/pmc_b.txt 2>&1 || exit 1
REPS=2 N0=$N0 rocprofv3 --pmc $PC --output-format csv -d $OUT/pmc_c -o r -- python tools/startup.py > $OUT/pmc_c.txt 2>&1 || exit 1
python tools/pmc_validation_summary.py $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $N0 2 > $OUT/pmc_valu_validation.json
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/startup.txt $OUT/gaps.txt
