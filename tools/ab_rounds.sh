#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for r in 2 3 4 5; do
  for rep in 1 2; do
    timeout -k 5 300 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --steps 10 --warmup 2 --rounds $r 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('rounds $r: bench value %.2f M  ms/step %.2f windows %d rounds %d trunc %d' % (o['value']/1e6, o['ms_per_step'], o['config']['windows_per_step'], o['config']['validation_rounds_per_step'], o['config']['truncated_windows_per_step']))" || exit 1
  done
done
