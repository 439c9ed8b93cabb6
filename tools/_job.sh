cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 2400 python -m pytest tests/test_gpu_stream.py tests/test_gpu_variants.py tests/test_gpu_wbuild.py tests/test_gpu_window.py tests/test_gpu_wunit.py tests/test_gpu_rank_device.py tests/test_gpu_rank_input.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error" | head -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
