#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kn in "chain_width=0" "chain_width=96" "chain_width=256"; do
  echo "== $kn"; timeout 900 python tests/perf_rank_input.py --knob $kn 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('pairs_per_pass','batches_per_pass','kind','sample_schedule_upload_s','train_s','train_pairs_per_s','end_to_end_pairs_per_s','overlapped_end_to_end_pairs_per_s') if k in d})"
done 2>&1 | tee gpurun_out/chain_rank_input.txt
