#!/bin/bash
R=/root/repo
cd $R
for c in "300 21 0 700" "132 21 0 700" "260 37 3 700" "5 37 0 700"; do
  timeout 120 python tools/debug_rank_tiles.py $c 2>&1 | grep -v "amdgpu.ids\|Extension\|^  File\|^$\|Thread" | tail -2 | cut -c1-300
done
timeout 900 python -m pytest tests/test_gpu_ranker.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for tk in 10 0; do
rm -rf /tmp/rk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk -o rk -- python $R/tests/perf_ranker.py --sections 1201 --cpu-sections 0 --top-k $tk > /tmp/rk.log 2>&1
f=$(find /tmp/rk -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r05_rank/kernel_stats_top$tk.csv
echo "== top_k $tk"
sed 's/^"[^"]*::\(k_[a-z_]*\)[^"]*"/\1/' $f | cut -c1-110 | grep "score_tile\|tile_select\|tile_open\|copyBuffer"
python $R/tests/perf_ranker.py --sections 1201 --cpu-sections 0 --top-k $tk 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gpu_bulk'])"
done
