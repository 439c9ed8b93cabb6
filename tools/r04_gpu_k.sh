#!/bin/bash
# round 4, call K: IPC exchange between processes on one GPU; basicMF one-rank share with bf16 contribution rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04k
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ipc.py -x -q > $OUT/ipc.log 2>&1
tail -15 $OUT/ipc.log
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "phase", {k: round(v, 2) for k, v in (d.get("phase_ms") or {}).items() if k != "what"}, "d", d.get("rmse_minus_sequential"))'
for c in fp32 bf16; do
  timeout 600 python bench.py --force-exchange --exchange minibatch --contrib $c --steps 5 --no-cpu-baseline --pmc off --secondary "" --no-window-step 2> $OUT/share_$c.log | python -c "$show" "one rank, full size, window-minibatch step, contributions $c" | tee -a $OUT/share.txt
done
