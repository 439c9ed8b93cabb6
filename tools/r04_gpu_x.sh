#!/bin/bash
# round 4, call X: SVD++ window step with bf16 contribution rows on seeds 0-2 (the one-GPU default now), and the 2-rank svdpp bench on one shared GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04x
WSTEP_CONTRIB=bf16 timeout 900 python tools/wstep_probe.py svdpp 0,1,2 16 > gpurun_out/r04x/probe_bf16.json 2> gpurun_out/r04x/probe_bf16.log
cut -c1-330 gpurun_out/r04x/probe_bf16.json
SVDF_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload svdpp --steps 2 --warmup 1 > gpurun_out/r04x/svdpp2.json 2> gpurun_out/r04x/svdpp2.log
tail -1 gpurun_out/r04x/svdpp2.json | cut -c1-1500
grep "\[bench\]" gpurun_out/r04x/svdpp2.log | tail -6 | cut -c1-250
