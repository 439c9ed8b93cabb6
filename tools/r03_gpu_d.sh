#!/bin/bash
# round 3, GPU call D: rank pairs through the window-minibatch step -- parity tests, one rank's share of BASELINE configs[4] at N = 8 / 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 600 python -X faulthandler -m pytest tests/test_gpu_window.py -x -q > $OUT/test_window.log 2>&1; echo "test_gpu_window rc=$?"; tail -3 $OUT/test_window.log
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "launches/pass", d["roofline"]["launches"]//d["steps"], {k: round(v,3) for k,v in p.items() if k!="what"}, "acc", d.get("pair_accuracy_test_after_run"))'
for n in 8 2; do
  for ex in minibatch levels; do
    timeout 900 python bench.py --workload pairwise --steps 2 --warmup 1 --no-cpu-baseline --force-exchange --exchange $ex --windows 125 --pairs $((200000000/n)) --users $((1000000/n)) --secondary "" 2>$OUT/pairs_${n}_$ex.log | python -c "$show" "pairs rank-of-$n $ex (125 windows)" | tee -a $OUT/pairs_probe.txt
  done
done
for kn in "window_slots=0"; do
  timeout 900 python bench.py --workload pairwise --steps 2 --warmup 1 --no-cpu-baseline --force-exchange --exchange minibatch --windows 125 --pairs 25000000 --users 125000 --secondary "" --knob $kn 2>/dev/null | python -c "$show" "pairs rank-of-8 minibatch $kn" | tee -a $OUT/pairs_probe.txt
done
