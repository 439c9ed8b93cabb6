#!/bin/bash
# round 4, call AA: neighbourhood window step, updates per global bias and window 48 / 64 / 96 at 10 passes on seeds 0-2 (contract 1e-4, target 8e-5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04aa
timeout 1700 python tools/wstep_probe.py neighbourhood 0,1,2 48,64,96 9 > gpurun_out/r04aa/probe.json 2> gpurun_out/r04aa/probe.log
cut -c1-290 gpurun_out/r04aa/probe.json
