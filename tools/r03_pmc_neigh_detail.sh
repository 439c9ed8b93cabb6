#!/bin/bash
# what the neighbourhood level's read traffic consists of: request sizes at the L2 <-> fabric interface and L2 request mix
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r03i}
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -oE "TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|SQC_[A-Z0-9_]+|SQ_INSTS_SMEM[A-Z_]*" | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
: > $OUT/pmc_neigh_detail.txt
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum" "TCC_RW_REQ_sum TCC_PROBE_sum" "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES" "TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum"; do
  n=$(echo $c | tr " " "_")
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/d_$n -o p -- python bench.py --workload neighbourhood --no-cpu-baseline --secondary "" --steps 1 --warmup 0 > /dev/null 2> $OUT/err_$n.log
  python tools/pmc_summary.py $OUT/d_$n 2>/dev/null | grep -E "k_fused" >> $OUT/pmc_neigh_detail.txt || echo "  ($c: not collected)" >> $OUT/pmc_neigh_detail.txt
  rm -rf $OUT/d_$n
done
cat $OUT/pmc_neigh_detail.txt
