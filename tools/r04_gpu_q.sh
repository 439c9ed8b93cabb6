#!/bin/bash
# round 4, call Q: pairs through the one-GPU window sequence (test + bench secondary), the drop-in opt-in test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_gpu_wunit.py tests/test_gpu_dropin_cli.py -x -q 2>&1 | tail -4
timeout 900 python bench.py --workload pairwise --secondary pairwise --pmc off --no-cpu-baseline --steps 2 > gpurun_out/r04q/pairs.json 2> gpurun_out/r04q/pairs.log
tail -3 gpurun_out/r04q/pairs.log | cut -c1-300
