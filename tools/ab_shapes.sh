#!/bin/bash
# A/B of k_scan_u against k_scan on the C5- and C4-shaped runs (tools/shapes.py); further build variants as arguments.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for su in 0 1; do
  export CHRONOCLUST_HIP_SCANU=$su
  echo "=== SCANU=$su"
  timeout -k 5 300 python tools/shapes.py ${SHAPES:-C5 C4} 2>&1 | grep -v amdgpu.ids || exit 1
done
for v in "$@"; do
  export CHRONOCLUST_HIP_LIB=$PWD/build/lib_$v.so
  echo "=== $v"
  timeout -k 5 300 python tools/shapes.py ${SHAPES:-C5 C4} 2>&1 | grep -v amdgpu.ids || exit 1
done
