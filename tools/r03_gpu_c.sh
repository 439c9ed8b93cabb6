#!/bin/bash
# round 3, GPU call C: native multi-GPU handle tests + the whole GPU suite + drop-in throughput with amd:gpus = 2 virtual ranks
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 300 python -X faulthandler -m pytest tests/test_gpu_native_multi.py -x -q > $OUT/native_multi.log 2>&1; echo "native_multi rc=$?"; tail -4 $OUT/native_multi.log
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -6 $OUT/gpu_suite.log
