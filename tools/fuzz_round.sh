#!/bin/bash
# extended randomised differential run (HIP engine vs C oracle), round-2 feature set; summary -> gpurun_out/fuzz_r02.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/fuzz_r02.txt
: > $out
for s in $(seq 101 124); do python tests/fuzz_parity.py --iters 1250 --seed $s 2>&1 | tail -1 | sed "s/^/seed $s: /" >> $out; done
for s in 201 202 203 204; do python tests/fuzz_parity.py --iters 400 --seed $s --big 2>&1 | tail -1 | sed "s/^/seed $s --big: /" >> $out; done
for s in 301 302; do python tests/fuzz_parity.py --iters 400 --seed $s --wide 2>&1 | tail -1 | sed "s/^/seed $s --wide: /" >> $out; done
python - <<PY >> $out
import json, re
tot = dict(iters=0, exact=0, tolerance=0, skipped=0)
for line in open("$out"):
    m = re.search(r"(\{.*\})", line)
    if m:
        d = json.loads(m.group(1))
        for k in tot: tot[k] += d.get(k, 0)
print("TOTAL", json.dumps(tot))
PY
tail -3 $out
