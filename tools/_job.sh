cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_auto_step.py tests/test_gpu_rank_input.py tests/test_gpu_rank_device.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|assert" | head
