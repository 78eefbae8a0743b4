#!/bin/bash
# Kernel experiments on the other shapes: tools/shapes.py (C4- / C5-shaped) for every build variant given.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for v in "$@"; do
  export CHRONOCLUST_HIP_LIB=$PWD/build/lib_$v.so
  echo "=== $v"
  timeout -k 5 300 python tools/shapes.py ${SHAPES:-C5 C4} 2>&1 | grep -v amdgpu.ids || exit 1
done
