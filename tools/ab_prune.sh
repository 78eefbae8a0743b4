#!/bin/bash
# A/B of the pruned snapshot scan (CHRONOCLUST_HIP_PRUNE = 0 plain k_scan_u, 1 auto, 2 always) on the GPU box:
# steady state alone (tools/steady.py LA=2), with lookahead, the bench headline, C4- and C5-shaped runs.
O=${1:-gpurun_out/ab_prune}
mkdir -p $O
for P in ${MODES:-0 1 2}; do
  export CHRONOCLUST_HIP_PRUNE=$P
  echo "== PRUNE=$P steady alone" ; LA=2 REPS=2 python tools/steady.py 2>&1 | grep steady
  echo "== PRUNE=$P steady lookahead" ; LA=0 REPS=2 python tools/steady.py 2>&1 | grep steady
  echo "== PRUNE=$P bench" ; python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'], o['roofline']['frac'], o['roofline']['avg_launch_us'])"
  echo "== PRUNE=$P shapes" ; python tools/shapes.py C4 C5 2>&1 | grep "N="
done
