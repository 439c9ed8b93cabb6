#!/bin/bash
# round 4, call R: one wave per user unit (k_wunit_wave) -- parity and A/B at the configs[3] SVD++ shape; the IPC flag page uncached
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04r
timeout 1200 python -m pytest tests/test_gpu_wunit.py tests/test_gpu_ipc.py -x -q 2>&1 | tail -6
for f in 1 2; do
  SVDF_WUNIT_FAST=$f timeout 600 python tools/wstep_probe.py svdpp 0 16 > gpurun_out/r04r/probe_fast$f.json 2> gpurun_out/r04r/probe_fast$f.log
  cat gpurun_out/r04r/probe_fast$f.json | cut -c1-400
done
timeout 900 python -m pytest tests/test_gpu_bench_multi.py -x -q 2>&1 | tail -4
