#!/bin/bash
# The CU mask of the lookahead-scan stream (CHRONOCLUST_HIP_SCAN_CUS; 0 = none): steady state and the bench headline.
for C in ${CUS:-0 240 224 208 192}; do
  echo "== CHRONOCLUST_HIP_SCAN_CUS=$C"
  CHRONOCLUST_HIP_SCAN_CUS=$C LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_SCAN_CUS=$C python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (b['value']/1e6, b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac']))"
done
