#!/bin/bash
# The tool outputs of a round that profiles/README.md lists beside profile_round.sh's (run on the GPU box from the repo root).
OUT=${1:-gpurun_out/extras}
mkdir -p $OUT
export TMPDIR=/tmp
{ echo "== LA=2 (scans alone)"; LA=2 REPS=2 python tools/steady.py 2>&1 | grep -A1 "steady run"; echo "== LA=0 (lookahead)"; LA=0 REPS=2 python tools/steady.py 2>&1 | grep -A1 "steady run";
  echo "== CHRONOCLUST_HIP_PRUNE=0, LA=2"; CHRONOCLUST_HIP_PRUNE=0 LA=2 REPS=2 python tools/steady.py 2>&1 | grep "steady run"; echo "== CHRONOCLUST_HIP_PRUNE=0, LA=0"; CHRONOCLUST_HIP_PRUNE=0 LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"; } > $OUT/tool_steady.txt
{ echo "== D=40 G=50000 N=2000000 LA=2"; N=2000000 D=40 G=50000 LA=2 REPS=1 python tools/steady.py 2>&1 | grep -A1 "steady run"; echo "== the same, CHRONOCLUST_HIP_PRUNE=0"; CHRONOCLUST_HIP_PRUNE=0 N=2000000 D=40 G=50000 LA=2 REPS=1 python tools/steady.py 2>&1 | grep "steady run"; } > $OUT/tool_steady_d40.txt
echo "steady done"
bash tools/ab_prune.sh $OUT > $OUT/tool_prune_ab.txt 2>&1
echo "ab done"
CHRONOCLUST_HIP_TRACE=1 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 1 --warmup 1 2>&1 >/dev/null | grep "^\[cc\]" | tail -30 > $OUT/tool_trace_c2.txt
python tools/c3.py > $OUT/tool_c3.txt 2>&1
python tools/c3_app.py > $OUT/tool_c3_app.txt 2>&1
bash tools/regime_grid.sh > $OUT/tool_regimes.txt 2>&1
echo "tools done"
python -m pytest tests/test_full_size_configs.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -20 > $OUT/tool_full_size.txt
ls $OUT
