#!/bin/bash
# round 4, final build: the GPU suite, smoke(), the builders' fuzz
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | grep -E "passed|failed|error|Error|FAILED" | tail -6 > gpurun_out/suite_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "amdgpu.ids" | tail -6 >> gpurun_out/suite_final.log
timeout 900 python tests/fuzz_builders.py --iters 200 --seed 9 2>&1 | tail -1 >> gpurun_out/suite_final.log
cat gpurun_out/suite_final.log
