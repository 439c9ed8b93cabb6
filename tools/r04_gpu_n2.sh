#!/bin/bash
# round 4: the driver's N = 2 command on one shared GPU (gloo), timed, with the new device-side builders
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SVDF_BENCH_SHARE_GPU=1
S=$(date +%s)
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/n2_line.json 2> gpurun_out/n2_stderr.log
E=$(date +%s)
echo "wall $((E-S)) s"
grep -E "^\[bench" gpurun_out/n2_stderr.log | cut -c1-220 | tail -40
tail -1 gpurun_out/n2_line.json | cut -c1-600
