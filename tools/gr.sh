#!/bin/bash
# Build container: rebuild the library if its sources changed (the loader refuses a stale one), then run a command on the GPU box.
#   tools/gr.sh <timeout seconds> '<command>'
python3 -c "from chronoclust_amd import build as b; b.build(); b.build_div_test()" || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
