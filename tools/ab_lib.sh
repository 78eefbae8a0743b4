#!/bin/bash
# A/B over build variants of the library (chronoclust_amd.build.build(out=..., defines=[...]); CHRONOCLUST_HIP_LIB):
#   tools/ab_lib.sh name1=path1.so name2=path2.so ...   steady state at C2's and C4's shapes, bench headline
for V in "$@"; do
  name=${V%%=*}; L=${V#*=}
  echo "== $name"
  CHRONOCLUST_HIP_LIB=$PWD/$L LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_LIB=$PWD/$L N=2000000 D=14 G=2000 LA=0 REPS=2 python tools/steady.py 2>&1 | grep "steady run"
  CHRONOCLUST_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (b['value']/1e6, b['ms_per_step'], b['roofline']['avg_launch_us'], b['roofline']['frac']))"
done
