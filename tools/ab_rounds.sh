for R in 2 3 4 5; do
  echo "== rounds $R"
  python bench.py --rounds $R --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.2f M  ms/step %.2f  windows %d rounds %d trunc %d' % (b['value']/1e6, b['ms_per_step'], b['config']['windows_per_step'], b['config']['validation_rounds_per_step'], b['config']['truncated_windows_per_step']))"
done
