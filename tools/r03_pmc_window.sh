#!/bin/bash
# PMC passes (one per counter group) of the window-minibatch kernels for ONE rank's share of an N-rank run (default N = 8):
#   tools/r03_pmc_window.sh [N] [tag]   -> gpurun_out/<tag>/pmc_window_rank_of_N.txt
set -u
N=${1:-8}; TAG=${2:-r03pmc}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
F=$OUT/pmc_window_rank_of_$N.txt
: > $F
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  n=$(echo $c | tr " " "_")
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcw_$n -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange minibatch --windows 32 --ratings $((100000000/N)) --users $((1000000/N)) --secondary "" > /dev/null 2> $OUT/pmcw_$n.stderr.log
  python tools/pmc_summary.py $OUT/pmcw_$n | grep -E "k_window|k_delta_addto|counter_collection" >> $F
  rm -rf $OUT/pmcw_$n $OUT/pmcw_$n.stderr.log
done
cat $F
