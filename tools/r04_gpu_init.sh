#!/bin/bash
# round 4: rand_init on the device -- tests, then the timing probe at the BASELINE shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_init.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/init_tests.log
cat gpurun_out/init_tests.log
timeout 600 python tools/init_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/init_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/init_parity.log
