#!/bin/bash
# PMC passes (one per counter group) for ONE bench workload: tools/pmc_one_workload.sh svdpp r02h  -> gpurun_out/<tag>/pmc_<workload>.txt
set -u
W=$1; TAG=${2:-r02}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/pmc_$W.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  n=$(echo $c | tr " " "_")
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$n -o p -- python bench.py --workload $W --no-cpu-baseline --secondary "" --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_${W}_$n.stderr.log
  python tools/pmc_summary.py $OUT/pmc_${W}_$n | grep -E "k_[a-z]+|counter_collection" >> $OUT/pmc_$W.txt
  rm -rf $OUT/pmc_${W}_$n $OUT/pmc_${W}_$n.stderr.log
done
cat $OUT/pmc_$W.txt
