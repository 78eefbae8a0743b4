#!/bin/bash
# Round 6: sub-ranges per point tile (partials per point) of the pruned scan, steady state alone (LA=2) and with lookahead.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for s in 64 32 16 8; do for la in 0 2; do echo "== SEG=$s LA=$la"; SEG=$s LA=$la REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [12]"; done; done
