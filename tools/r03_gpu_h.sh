#!/bin/bash
# round 3, GPU call H: full GPU suite, then the profile set of the round on the same build (bench line, kernel stats, PMC)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03h
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; grep -E "passed|failed" $OUT/gpu_suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 2400 bash tools/profile_round3.sh r03h_prof > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
