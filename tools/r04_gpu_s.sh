#!/bin/bash
# round 4, call S: kernel trace of the SVD++ window step with one wave per unit (fp32 and bf16 contribution rows)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_wunit.py -x -q 2>&1 | tail -4
for c in fp32 bf16; do
WSTEP_CONTRIB=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04s/prof_$c -o p -- python tools/wstep_probe.py svdpp 0 16 > gpurun_out/r04s/probe_$c.json 2> gpurun_out/r04s/probe_$c.log
cat gpurun_out/r04s/probe_$c.json | cut -c1-300
f=$(find gpurun_out/r04s/prof_$c -name '*kernel_stats.csv' | head -1)
grep wunit "$f" | cut -c1-200
done
find gpurun_out/r04s -name '*.db' -delete; find gpurun_out/r04s -name '*kernel_trace.csv' -delete
