#!/bin/bash
# Window size of the steady state (tuning `window`; the library caps it at 32 768): steady state and the bench headline.
for W in ${WINS:-16384 24576 28672 32768}; do
  echo "== window $W"
  WIN=$W LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  python bench.py --window $W --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (b['value']/1e6, b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac']))"
done
