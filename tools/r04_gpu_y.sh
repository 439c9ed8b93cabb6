#!/bin/bash
# round 4, call Y: single contributions applied in place (one-GPU window sequences): parity, then A/B at the configs[3] sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04y
timeout 1200 python -m pytest tests/test_gpu_wunit.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
for W in neighbourhood svdpp; do
  timeout 600 python tools/wstep_probe.py $W 0 $([ $W = svdpp ] && echo 16 || echo 24) > gpurun_out/r04y/probe_$W.json 2> gpurun_out/r04y/probe_$W.log
  cut -c1-300 gpurun_out/r04y/probe_$W.json
done
