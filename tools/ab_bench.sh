#!/bin/bash
# Bench line (10 steps) for combinations of the validation kernels' workgroup sizes: tools/ab_bench.sh "chain decide commit" ...
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for cfg in "$@"; do
  set -- $cfg
  export CHRONOCLUST_HIP_CHAIN_THREADS=$1 CHRONOCLUST_HIP_DECIDE_THREADS=$2 CHRONOCLUST_HIP_COMMIT_THREADS=$3
  for rep in 1 2; do
    timeout -k 5 300 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('chain/decide/commit $cfg: bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (o['value']/1e6, o['ms_per_step'], o['roofline']['avg_launch_us'], o['roofline']['frac']))" || exit 1
  done
done
