#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_rccl_native.py tests/test_gpu_init.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -15 | tee gpurun_out/rccl_tests.log
