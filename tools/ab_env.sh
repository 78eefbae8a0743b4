#!/bin/bash
# A/B of one environment switch of the library: tools/ab_env.sh NAME value1 value2 ... (steady state alone / lookahead, bench)
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
name=$1; shift
for v in "$@"; do
  export $name=$v
  echo "=== $name=$v"
  LA=2 REPS=1 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
  LA=0 REPS=2 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
  timeout -k 5 300 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (o['value']/1e6, o['ms_per_step'], o['roofline']['avg_launch_us'], o['roofline']['frac']))" || exit 1
done
