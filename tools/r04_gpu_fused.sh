#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_window.py tests/test_gpu_rccl_native.py tests/test_gpu_ipc.py tests/test_gpu_bench_multi.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -12 | tee gpurun_out/fused_tests.log
timeout 600 python tools/native_ring_probe.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/native_ring_probe.txt
