#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_wunit.py tests/test_gpu_window.py tests/test_gpu_ipc.py -x -q 2>&1 | tail -3
rm -f gpurun_out/r04o/shares.txt
bash tools/r04_gpu_o.sh
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "phase_ms", {k: round(v,3) for k,v in p.items() if k!="what"})'
python bench.py --force-exchange --exchange minibatch --contrib bf16 --steps 5 --no-cpu-baseline --pmc off --secondary "" --no-window-step --no-sequential-reference 2>/dev/null | python -c "$show" "one rank, full size, all-reduce step, bf16" | tee -a gpurun_out/r04o/shares.txt
WSTEP_CONTRIB=bf16 python tools/wstep_probe.py svdpp 0 16 2>/dev/null | cut -c1-200
