#!/bin/bash
# probes of the pruned chain on / off: C2 headline, the C5-shaped one-stream leg, C3's later timepoints (same box)
for p in 1 0 1 0; do
  CHRONOCLUST_HIP_PROBE=$p python bench.py --no-cpu-baseline --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=o['one_stream_exact']; print('PROBE=$p: C2 %.2f ms/step %.1f M/s | C5-shaped leg %.2f M/s %.0f ms' % (o['ms_per_step'], o['value']/1e6, l['value']/1e6, l['ms_per_step']))"
done
for p in 1 0; do echo "PROBE=$p"; CHRONOCLUST_HIP_PROBE=$p python tools/c3.py 2>&1 | grep "^t=" | cut -c1-80; done
