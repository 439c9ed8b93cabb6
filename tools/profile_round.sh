#!/bin/bash
# Run on the GPU box through gpurun:  tools/profile_round.sh r05
# gpurun_out/<tag>/: the default bench line (N = 1: main + secondaries + the opt-in window-step lines, traffic measured in the run), rocprofv3 kernel
# stats of the same command (SVD++ at 40 K users and without secondary.orders: their deep exact passes are millions of launches), and PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) of the user-unit window kernels.
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.stderr.log
tail -1 $OUT/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --pmc off --no-orders --svdpp-users 40000 > $OUT/kt_bench.json 2> $OUT/kt.stderr.log
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
head -16 $OUT/kernel_stats.csv | cut -c1-200
for W in svdpp neighbourhood; do
  : > $OUT/pmc_wstep_$W.txt
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
    n=$(echo $c | tr " " "_")
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$n -o p -- python tools/wstep_probe.py $W 0 $([ $W = svdpp ] && echo 16 || echo 24) 1 > /dev/null 2> $OUT/pmc_${W}_$n.stderr.log
    python tools/pmc_summary.py $OUT/pmc_${W}_$n | grep -E "k_wunit|counter_collection" >> $OUT/pmc_wstep_$W.txt
    rm -rf $OUT/pmc_${W}_$n $OUT/pmc_${W}_$n.stderr.log
  done
  echo "== window step, $W"; cat $OUT/pmc_wstep_$W.txt
done
