#!/bin/bash
# Knobs on the start-up stretch (tools/startup.py): one line per setting (the last repetition).
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
run() { echo -n "$* : "; env "$@" REPS=4 python tools/startup.py 2>&1 | tail -1; }
run X=0
run CHRONOCLUST_HIP_DSCANU=0
