#!/bin/bash
# rocprofv3 kernel stats + FETCH/WRITE of ONE rank's share of the stratified schedule on 8 ranks (64 steps per pass, two item blocks per rank)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03x
mkdir -p $OUT
n=8
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange stratified --chunks 64 --ratings $((100000000/n)) --users $((1000000/n)) --items $((100000/(2*n))) --secondary ''"
eval rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > /dev/null 2> $OUT/kt.log
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/stratified_rank_of_8_kernel_stats.csv \; ; rm -rf $OUT/kt
head -8 $OUT/stratified_rank_of_8_kernel_stats.csv | cut -c1-160
: > $OUT/stratified_rank_of_8_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  eval rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_$c -o p -- $CMD > /dev/null 2> $OUT/p.log
  python tools/pmc_summary.py $OUT/p_$c | grep -E "k_window|k_ranges" >> $OUT/stratified_rank_of_8_pmc.txt
  rm -rf $OUT/p_$c
done
cat $OUT/stratified_rank_of_8_pmc.txt
