#!/bin/bash
# Run on the GPU box through gpurun:  tools/profile_round3.sh r03
# Produces gpurun_out/<tag>/ : the bench line, rocprofv3 kernel stats of the same command (main line + secondary workloads),
# and per-workload PMC summaries (FETCH/WRITE sizes, TCC EA requests, SQ instruction counts) of its dominant kernel.
# Counters are collected in their own passes (--kernel-trace + --pmc only, never with sys/hip/hsa tracing).
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.stderr.log
tail -1 $OUT/bench.json | cut -c1-300
# per-kernel time of the same command (all workloads of the default line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt.stderr.log
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
head -12 $OUT/kernel_stats.csv
# hardware counters: one pass per counter group and workload, 1 pass over the data, no warm-up (tools/pmc_one_workload.sh = the same for one workload)
for W in basicmf pairwise svdpp neighbourhood; do
  : > $OUT/pmc_$W.txt
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    n=$(echo $c | tr " " "_")
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$n -o p -- python bench.py --workload $W --no-cpu-baseline --secondary "" --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_${W}_$n.stderr.log
    python tools/pmc_summary.py $OUT/pmc_${W}_$n | grep -E "k_[a-z]+|counter_collection" >> $OUT/pmc_$W.txt
    rm -rf $OUT/pmc_${W}_$n $OUT/pmc_${W}_$n.stderr.log
  done
  echo "== $W"; cat $OUT/pmc_$W.txt
done
python tools/pmc_to_json.py $OUT $OUT/hbm_traffic.json
