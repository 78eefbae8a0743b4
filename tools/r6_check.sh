#!/bin/bash
# Round 6: quick check of a scan change - the pruned-scan tests, the steady state, the start-up stretch, one bench line.
set -o pipefail
OUT=${1:-gpurun_out/r6check}
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
timeout -k 10 600 python3 -m pytest tests/test_pruned_scan.py tests/test_code_path_knobs.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [12]"
REPS=3 python3 tools/startup.py 2>&1 | grep "run [12]"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs 2>/dev/null | cut -c1-140
CHRONOCLUST_HIP_TRACE=1 python3 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs 2>&1 | grep "offline phase" | tail -1
