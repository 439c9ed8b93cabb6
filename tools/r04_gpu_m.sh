#!/bin/bash
# round 4, call M: the whole GPU suite on the current build, then the profile script
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04m
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r04m/gpu_suite.log 2>&1
tail -6 gpurun_out/r04m/gpu_suite.log
timeout 2400 bash tools/profile_round4.sh r04
