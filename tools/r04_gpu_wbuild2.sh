#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_wbuild.py tests/test_gpu_init.py tests/test_gpu_window.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -15 | tee gpurun_out/wbuild_tests.log
timeout 900 python tools/wbuild_probe.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/wbuild_probe.txt
for s in 2 3; do timeout 1200 python tests/fuzz_builders.py --iters 300 --seed $s 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3 | sed "s/^/builders seed $s: /"; done | tee gpurun_out/fuzz_builders2.txt
