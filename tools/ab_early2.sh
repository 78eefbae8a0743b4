mkdir -p gpurun_out/r04d
for ew in 4096 8192 16384; do
  python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 --early-window $ew > gpurun_out/r04d/bench_ew$ew.json 2> gpurun_out/r04d/bench_ew$ew.err
  python -c "
import json,sys
o=json.loads(open('gpurun_out/r04d/bench_ew$ew.json').read().strip().splitlines()[-1]); print('early-window $ew: %.2f ms/step %.1f M/s windows %d rounds %d trunc %d' % (o['ms_per_step'], o['value']/1e6, o['config']['windows_per_step'], o['config']['validation_rounds_per_step'], o['config']['truncated_windows_per_step']))"
done
CHRONOCLUST_HIP_TRACE=1 python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 1 --warmup 1 --early-window 16384 2>&1 >/dev/null | grep "^\[cc\]" | tail -14 | cut -c1-250
