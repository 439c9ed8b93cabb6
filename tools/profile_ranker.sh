#!/bin/bash
# ranker throughput (both output modes) + rocprofv3 kernel stats of the positions mode; results under gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
python $R/tests/perf_ranker.py 2>/dev/null | tail -1 > $R/gpurun_out/ranker_positions.json
python $R/tests/perf_ranker.py --top-k 10 2>/dev/null | tail -1 > $R/gpurun_out/ranker_topk10.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk -o rk -- python $R/tests/perf_ranker.py --cpu-sections 0 > /tmp/rk.log 2>&1
f=$(find /tmp/rk -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/ranker_kernel_stats.csv
rm -rf /tmp/rk2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk2 -o rk -- python $R/tests/perf_ranker.py --cpu-sections 0 --top-k 10 > /tmp/rk2.log 2>&1
f=$(find /tmp/rk2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/ranker_topk_kernel_stats.csv
cat $R/gpurun_out/ranker_positions.json $R/gpurun_out/ranker_topk10.json
head -7 $R/gpurun_out/ranker_kernel_stats.csv
head -12 $R/gpurun_out/ranker_topk_kernel_stats.csv
tail -3 /tmp/rk.log
