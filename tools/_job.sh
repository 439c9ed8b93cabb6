cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05a
timeout 300 tools/dag_probe/dag_probe > gpurun_out/r05a/dag_probe.txt 2>&1; echo "probe rc $?" >> gpurun_out/r05a/dag_probe.txt
timeout 1500 python -m pytest tests/test_gpu_auto_step.py tests/test_gpu_ipc.py tests/test_gpu_native_multi.py tests/test_gpu_wunit.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r05a/pytest.log
cat gpurun_out/r05a/dag_probe.txt; cat gpurun_out/r05a/pytest.log
