#!/bin/bash
# A/B of round 4: partials per point of the pruned scans (CHRONOCLUST_HIP_PRUNE_WGS), the skewed-population stream
B="python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1"
for w in 4 8 16 32; do
  CHRONOCLUST_HIP_PRUNE_WGS=$w $B 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRUNE_WGS=$w: %.2f ms/step %.1f M/s' % (o['ms_per_step'], o['value']/1e6))"
  CHRONOCLUST_HIP_PRUNE_WGS=$w LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run 1"
done
python tools/skewed.py
