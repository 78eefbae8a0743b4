#!/bin/bash
# Round 6: seeds of the seeded chains from the matrix cores (k_seed16 + tight threshold) or from k_seed (F = 16): the general regimes of the bench
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for v in 1 0 1 0; do echo "== SEED16=$v"; CHRONOCLUST_HIP_SEED16=$v python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs 2>/dev/null | cut -c1-130; python3 -c "
import json
d=json.load(open(\"bench_detail.json\"))
print({k:(round(v[\"value\"]/1e6,2), round(v[\"ms_per_step\"],2), v[\"pruned_scans_per_step\"], v[\"truncated_windows_per_step\"]) for k,v in d[\"general_regimes\"].items()})
"; done
