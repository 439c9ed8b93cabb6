#!/bin/bash
# round 4, call U: the full GPU suite and the round's profiles on the build with one wave per user unit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04u
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04u/gpu_suite.log
cat gpurun_out/r04u/gpu_suite.log
timeout 1800 bash tools/profile_round4.sh r04u 2>&1 | tail -40 | cut -c1-250
