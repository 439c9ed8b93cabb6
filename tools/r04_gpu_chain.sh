#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_sched.py tests/test_gpu_rank_device.py tests/test_gpu_rank_input.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -12 | tee gpurun_out/chain_tests.log
timeout 600 python tools/chain_probe.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/chain_probe.txt
