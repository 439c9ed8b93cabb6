#!/bin/bash
# N>1 bench plumbing on a ONE-GPU box: every rank on GPU 0, exchange through gloo (SVDF_BENCH_SHARE_GPU=1).  Exercises the
# self-spawn of bench.py --gpus N (no torch.distributed.run on the command line), sharding, windows, exchange, timing, the
# quality reduction and the JSON line for the three sharded workloads.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SVDF_BENCH_SHARE_GPU=1
python bench.py --gpus 2 --ratings 10000000 --cpu-sample 2000000 --steps 2 2> gpurun_out/multi_basicmf.log | cut -c1-700
python bench.py --gpus 2 --workload pairwise --pairs 10000000 --cpu-sample 2000000 --steps 2 2> gpurun_out/multi_pairwise.log | cut -c1-700
python bench.py --gpus 3 --workload svdpp --svdpp-users 6000 --cpu-sample 2000000 --steps 2 2> gpurun_out/multi_svdpp.log | cut -c1-700
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --ratings 5000000 --no-cpu-baseline --steps 2 2>> gpurun_out/multi_basicmf.log | cut -c1-300
tail -3 gpurun_out/multi_basicmf.log gpurun_out/multi_pairwise.log gpurun_out/multi_svdpp.log
