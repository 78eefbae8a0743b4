#!/bin/bash
# Partials per point of the snapshot scan on the C5-shaped run (SEG = sub-ranges per point tile = 4 x partials at most).
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for v in 24 32 48 64 96; do
  echo "=== segments $v"
  SEG=$v timeout -k 5 300 python tools/shapes.py C5 2>&1 | grep -v amdgpu.ids || exit 1
done
