#!/bin/bash
# Sub-range split of the pruned scan (CHRONOCLUST_HIP_PRUNE_WGS: workgroups per CU the chain's grids are sized for)
for W in ${SPLITS:-8 12 16 24}; do
  echo "== CHRONOCLUST_HIP_PRUNE_WGS=$W"
  CHRONOCLUST_HIP_PRUNE_WGS=$W LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_PRUNE_WGS=$W N=2000000 D=40 G=50000 LA=2 REPS=1 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_PRUNE_WGS=$W N=2000000 D=14 G=2000 LA=0 REPS=1 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_PRUNE_WGS=$W python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (b['value']/1e6, b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac']))"
done
