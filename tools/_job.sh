cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05f


timeout 1500 python -m pytest tests/test_gpu_native_multi.py tests/test_gpu_dropin_cli.py tests/test_gpu_window.py tests/test_gpu_bench_multi.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5
