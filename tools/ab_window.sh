#!/bin/bash
# Round 6: the default window (49 152 since round 6; the policy holds 32 768 on small tables) at the shapes of README.md; WIN=32768: before
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for w in 0 32768; do
echo "== WIN=$w: C2 steady"; WIN=$w REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [2]"
echo "== WIN=$w: C5 shape steady"; WIN=$w D=40 G=50000 N=2000000 REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]"
echo "== WIN=$w: C4 shape steady"; WIN=$w D=14 G=2000 N=2000000 REPS=2 python3 tools/steady.py 2>&1 | grep "steady run [1]"
echo "== WIN=$w: skewed"; WIN=$w N=2000000 D=14 G=2000 HEAVY=0.3 python3 tools/skewed.py 2>&1 | tail -1
echo "== WIN=$w: start-up"; WIN=$w REPS=3 python3 tools/startup.py 2>&1 | grep "run [2]"
echo "== WIN=$w: bench"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --window $w --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs 2>/dev/null | cut -c1-140
done
