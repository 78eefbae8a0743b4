#!/bin/bash
# the sparse dirty scans' threshold (one flagged point in N): bench headline per setting
for n in 0 32 128 512; do
  CHRONOCLUST_HIP_SPARSE=$n python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CHRONOCLUST_HIP_SPARSE=$n: %.2f ms/step %.1f M/s windows %d lookahead %d rounds %d trunc %d' % (o['ms_per_step'], o['value']/1e6, o['config']['windows_per_step'], o['config']['lookahead_windows_per_step'], o['config']['validation_rounds_per_step'], o['config']['truncated_windows_per_step']))"
done
