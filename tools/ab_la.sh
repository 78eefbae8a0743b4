#!/bin/bash
# Round 6: with the snapshot scan at a quarter of its former time, do lookahead scans still pay?  bench (C2) and the C4-shaped stream
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs"
for i in 1 2; do
echo "== bench default"; $B 2>/dev/null | cut -c1-140
echo "== bench --lookahead 2"; $B --lookahead 2 2>/dev/null | cut -c1-140
done
for la in 0 2; do
echo "== C4 shape (2 M x 14, 2000 MCs) LA=$la"; D=14 G=2000 N=2000000 LA=$la REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [12]"
done
