#!/bin/bash
# Round 6: the start-up stretch with lookahead scans forced on from the first window (LA=3) against the default policy.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for la in 0 3; do echo "== LA=$la"; LA=$la CHRONOCLUST_HIP_TRACE=1 REPS=2 python3 tools/startup.py 2>&1 | tail -16 | cut -c1-330; done
