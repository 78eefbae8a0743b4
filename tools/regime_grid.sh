#!/bin/bash
# few-microcluster regimes (tools/one_regime.py: 500 k points, first / second run of a process) and the bundled d0-d4 data
for D in 20 5; do for G in 12 50 200 1000; do D=$D G=$G REPS=2 python tools/one_regime.py 2>&1 | grep "blobs"; done; done
for G in 30 100 300; do N=2000000 D=14 G=$G REPS=2 python tools/one_regime.py 2>&1 | grep "blobs"; done
python tests/extra/gpu_debug.py c1 2>&1 | grep "t=" | head -5
