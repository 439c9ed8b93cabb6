#!/bin/bash
# round 4, call V: the full GPU suite (log kept)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04v
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04v/gpu_suite.log 2>&1
grep -E "passed|failed|error" gpurun_out/r04v/gpu_suite.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
