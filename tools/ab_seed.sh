#!/bin/bash
# k_seed against k_seed2 (CHRONOCLUST_HIP_SEED=0 / 1 / 2): the scan alone, the steady state, the bench headline.
for M in ${MODES:-0 1 2}; do
  echo "== CHRONOCLUST_HIP_SEED=$M"
  CHRONOCLUST_HIP_SEED=$M WIN=${WIN:-0} LA=2 REPS=1 python tools/steady.py 2>&1 | grep -A1 "steady run"
  CHRONOCLUST_HIP_SEED=$M WIN=${WIN:-0} LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_SEED=$M N=2000000 D=40 G=50000 LA=2 REPS=1 python tools/steady.py 2>&1 | grep -A1 "steady run"
  CHRONOCLUST_HIP_SEED=$M python bench.py --window ${WIN:-0} --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (b['value']/1e6, b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac']))"
done
