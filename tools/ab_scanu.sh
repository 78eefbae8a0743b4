#!/bin/bash
# A/B of k_scan_u against k_scan: steady state (scans alone, then with lookahead), the bench line, the long steady run.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for su in 0 1; do
  export CHRONOCLUST_HIP_SCANU=$su
  echo "=== SCANU=$su"
  N=4000000 LA=2 REPS=1 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
  N=4000000 LA=0 REPS=1 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
  timeout -k 5 300 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --steps 3 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (o['value']/1e6, o['ms_per_step'], o['roofline']['avg_launch_us'], o['roofline']['frac']))" || exit 1
done
