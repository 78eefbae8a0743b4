#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for v in 16 32 64 128 256; do
  export DSEG=$v
  for rep in 1 2; do
    timeout -k 5 100 python tools/run_c2.py 2>&1 | grep "^{" | python -c "import sys,ast; o=ast.literal_eval(sys.stdin.read()); print('dirty segments $v: run %.2f ms windows %d rounds %d trunc %d' % (o['run_ms'], o['windows'], o['rounds'], o['truncated']))" || exit 1
  done
done
