#!/bin/bash
# round 4, call AC: wall time of the default N = 2 bench command (shared GPU + gloo: an upper bound for two real devices)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04ac
t0=$(date +%s)
SVDF_BENCH_SHARE_GPU=1 timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 > gpurun_out/r04ac/n2.json 2> gpurun_out/r04ac/n2.log
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -1 gpurun_out/r04ac/n2.json | cut -c1-600
grep "\[bench\]" gpurun_out/r04ac/n2.log | cut -c1-200 | tail -40
