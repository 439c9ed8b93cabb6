#!/bin/bash
# The GPU-box jobs of a round, one parametrised script (run through gpurun):
#   tools/gpu_round.sh suite   [TAG]            GPU test suite + smoke()
#   tools/gpu_round.sh fuzz    [TAG] [SEED0]    randomised differential runs (parity / wunit / multi / ranker / builders)
#   tools/gpu_round.sh profile [TAG]            tools/profile_round.sh TAG: bench line, rocprofv3 kernel stats, PMC of the window kernels
#   tools/gpu_round.sh pytest  [TAG] ARGS...    a pytest selection, output to gpurun_out/TAG/pytest.log
# Everything lands under gpurun_out/TAG/ (scratch; copy what is to be judged into profiles/).
set -u
JOB=${1:-suite}; TAG=${2:-r05}; shift 2 2>/dev/null
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG; mkdir -p $OUT
nolog() { grep -v "amdgpu.ids"; }
case $JOB in
suite)
  python -m pytest tests -m gpu -x -q 2>&1 | nolog | grep -E "passed|failed|error|Error|FAILED" | tail -8 > $OUT/suite.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | nolog | tail -6 >> $OUT/suite.log
  cat $OUT/suite.log ;;
pytest)
  python -m pytest "$@" 2>&1 | nolog | tail -40 > $OUT/pytest.log; cat $OUT/pytest.log ;;
fuzz)
  S=${1:-6000}; out=$OUT/fuzz.txt; : > $out
  for s in $(seq $S $((S+7))); do timeout 900 python tests/fuzz_parity.py --iters 1250 --seed $s 2>&1 | tail -1 | sed "s/^/parity seed $s: /" >> $out; done
  timeout 900 python tests/fuzz_parity.py --iters 400 --seed $((S+50)) --big 2>&1 | tail -1 | sed "s/^/parity --big: /" >> $out
  for s in $((S+60)) $((S+61)); do timeout 900 python tests/fuzz_wunit.py --iters 400 --seed $s 2>&1 | tail -2 | sed "s/^/wunit seed $s: /" >> $out; done
  timeout 900 python tests/fuzz_multi.py --iters 800 --seed $((S+70)) 2>&1 | tail -1 | sed "s/^/multi: /" >> $out
  timeout 900 python tests/fuzz_ranker.py --iters 1000 --seed $((S+80)) 2>&1 | tail -2 | sed "s/^/ranker: /" >> $out
  timeout 900 python tests/fuzz_builders.py --iters 200 --seed $((S+90)) 2>&1 | tail -1 | sed "s/^/builders: /" >> $out
  echo "MISMATCH lines: $(grep -c MISMATCH $out)"; cut -c1-220 $out ;;
profile)
  bash tools/profile_round.sh $TAG ;;
*) echo "unknown job $JOB"; exit 2 ;;
esac
