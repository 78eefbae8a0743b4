#!/bin/bash
# working tree against a build of the last commit (chronoclust_amd/libcc_head.so, see tools/README.md), same box, interleaved
B="python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1"
for i in 1 2 3; do
  for v in head tree; do
    if [ $v = head ]; then export CHRONOCLUST_HIP_LIB=$PWD/chronoclust_amd/libcc_head.so; else unset CHRONOCLUST_HIP_LIB; fi
    $B 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.2f ms/step %.1f M/s' % (o['ms_per_step'], o['value']/1e6))"
  done
done
