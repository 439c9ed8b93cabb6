#!/bin/bash
# round 4, call T: PMC counters of the one-wave-per-unit window kernel (what bounds it: traffic, issue, waiting)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04t_${W:-svdpp}; mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/pmc_wave.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python tools/wstep_probe.py ${W:-svdpp} 0 ${PT:-16} 1 > /dev/null 2> $OUT/pmc_$n.stderr.log
  python tools/pmc_summary.py $OUT/pmc_$n | grep -E "k_wunit|counter_collection" >> $OUT/pmc_wave.txt
  tail -2 $OUT/pmc_$n.stderr.log | cut -c1-200
  rm -rf $OUT/pmc_$n
done
cat $OUT/pmc_wave.txt
