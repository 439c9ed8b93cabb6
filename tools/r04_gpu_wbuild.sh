#!/bin/bash
# round 4: window data sets regrouped on the device -- tests, then the build-time probe at the BASELINE sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_wbuild.py tests/test_gpu_window.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -15 | tee gpurun_out/wbuild_tests.log
timeout 900 python tools/wbuild_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/wbuild_probe.txt
