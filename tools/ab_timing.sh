#!/bin/bash
# What the per-launch HIP events around the snapshot scans cost the bench step (10 steps each, twice).
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for flag in "" "--no-kernel-timing"; do
  for rep in 1 2; do
    timeout -k 5 300 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --steps 10 --warmup 2 $flag 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('flag [$flag]: bench value %.2f M  ms/step %.2f' % (o['value']/1e6, o['ms_per_step']))" || exit 1
  done
done
