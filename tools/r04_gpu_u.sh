#!/bin/bash
# round 4, call U: the round's profiles on the final build (tools/profile_round4.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04u
timeout 1800 bash tools/profile_round4.sh r04u 2>&1 | tail -45 | cut -c1-250
