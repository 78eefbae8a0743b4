#!/bin/bash
# End-of-round tool runs (GPU box): C3 pipeline, other shapes, few-MC regimes, bundled data.  bash tools/round_tools.sh <outdir>
OUT=${1:-gpurun_out/r02_tools}
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
N=1000000 timeout -k 5 300 python tools/c3.py 2>&1 | grep -v amdgpu > $OUT/c3.txt || exit 1
timeout -k 5 300 python tools/shapes.py C4 C5 2>&1 | grep -v amdgpu > $OUT/shapes.txt || exit 1
(for D in 20 5; do for G in 12 50 200 1000; do D=$D G=$G N=500000 timeout -k 5 120 python tools/one_regime.py || exit 1; done; done) 2>&1 | grep blobs > $OUT/regimes.txt
(WINDOW=0 timeout -k 5 100 python tests/extra/gpu_debug.py c1; echo "--- sequential kernel off:"; SEQUENTIAL=1 WINDOW=0 timeout -k 5 100 python tests/extra/gpu_debug.py c1) 2>&1 | grep "t=\|---" > $OUT/c1.txt
timeout -k 5 200 python tools/offline_prof.py 2>&1 | grep offline > $OUT/offline.txt
cat $OUT/c3.txt $OUT/shapes.txt $OUT/regimes.txt $OUT/c1.txt $OUT/offline.txt
