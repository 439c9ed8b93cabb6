#!/bin/bash
# builds nothing (the binary is built here by tools/pmc_calib/build.sh and travels with the snapshot); counters in separate passes
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r03_calib}
mkdir -p $OUT
BIN=tools/pmc_calib/pmc_calib
timeout 120 $BIN > $OUT/expected.txt 2>&1; cat $OUT/expected.txt
: > $OUT/pmc_calib.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr " " "_")
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/d_$n -o p -- $BIN > /dev/null 2> $OUT/err_$n.log
  python tools/pmc_summary.py $OUT/d_$n 2>/dev/null | grep -vE "counter_collection|^$" >> $OUT/pmc_calib.txt
  rm -rf $OUT/d_$n
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BIN > /dev/null 2> $OUT/err_kt.log
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; ; rm -rf $OUT/kt
cat $OUT/pmc_calib.txt; cut -d, -f1-4 $OUT/kernel_stats.csv
