#!/bin/bash
# round 4, call G: new tests of the ADVICE fixes + the contract over data seeds at the full configs[2] size
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04g
timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_native_multi.py -x -q > gpurun_out/r04g/tests.log 2>&1
tail -5 gpurun_out/r04g/tests.log
timeout 3000 python tools/contract_seeds.py 0,1,2 2,4,8 > gpurun_out/r04g/contract.jsonl 2> gpurun_out/r04g/contract.log
cat gpurun_out/r04g/contract.jsonl; tail -5 gpurun_out/r04g/contract.log
