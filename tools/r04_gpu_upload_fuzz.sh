#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 tools/upload_probe/upload_probe 2>&1 | tee gpurun_out/upload_probe.txt
for s in 1 2; do timeout 1200 python tests/fuzz_builders.py --iters 300 --seed $s 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 | sed "s/^/builders seed $s: /"; done | tee gpurun_out/fuzz_builders.txt
