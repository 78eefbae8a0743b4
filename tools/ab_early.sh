#!/bin/bash
# Window of the start-up phase (tuning early_window) and validation rounds: the bench headline and its trace.
for E in ${EARLY:-2048 4096 6144 8192 16384}; do
  echo "== early_window $E"
  python bench.py --early-window $E --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (b['value']/1e6, b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac']))"
  CHRONOCLUST_HIP_TRACE=1 python bench.py --early-window $E --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 1 --warmup 1 2>&1 >/dev/null | grep "^\[cc\]" | grep -v pruned | tail -14 | cut -c1-150
done
