#!/bin/bash
# Scan workgroup shape experiments: build variant x segments.  Usage (GPU box): bash tools/variants_nw.sh
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
run() {
  export CHRONOCLUST_HIP_LIB=$PWD/build/lib_$1.so
  echo "=== $1 segments $2"
  SEG=$2 LA=2 REPS=1 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
  SEG=$2 LA=0 REPS=1 timeout -k 5 200 python tools/steady.py 2>&1 | grep "steady run" || exit 1
  timeout -k 5 300 python bench.py --no-cpu-baseline --no-one-stream --steps 3 --segments $2 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('bench value %.2f M  ms/step %.2f  scan avg %.1f us frac %.3f' % (o['value']/1e6, o['ms_per_step'], o['roofline']['avg_launch_us'], o['roofline']['frac']))" || exit 1
}
run base3 64
run nw8 64
run nw8 32
run nw16 64
run nw16 32
