#!/bin/bash
# randomised differential run of round 3 (the round's last build: streaming model save, zero-copy predictions, handle entry points): training paths (fresh seeds) + ranker streams;
# summary -> gpurun_out/fuzz_r03d.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/fuzz_r03d.txt
: > $out
for s in $(seq 3211 3230); do python tests/fuzz_parity.py --iters 1250 --seed $s 2>&1 | tail -1 | sed "s/^/seed $s: /" >> $out; done
for s in 3251 3252; do python tests/fuzz_parity.py --iters 400 --seed $s --big 2>&1 | tail -1 | sed "s/^/seed $s --big: /" >> $out; done
for s in 3261; do python tests/fuzz_parity.py --iters 400 --seed $s --wide 2>&1 | tail -1 | sed "s/^/seed $s --wide: /" >> $out; done
for s in $(seq 3271 3273); do python tests/fuzz_ranker.py --iters 1000 --seed $s 2>&1 | tail -3 | sed "s/^/ranker seed $s: /" >> $out; done
for s in 21 22; do python tests/fuzz_multi.py --iters 800 --seed $s 2>&1 | tail -1 | sed "s/^/multi seed $s: /" >> $out; done
python - <<PY >> $out
import json, re
tot = dict(iters=0, exact=0, tolerance=0, skipped=0, failed=0)
for line in open("$out"):
    m = re.search(r"(\{.*\})", line)
    if m and "MISMATCH" not in line:
        d = json.loads(m.group(1))
        for k in tot: tot[k] += d.get(k, 0)
print("TOTAL", json.dumps(tot))
PY
tail -4 $out
