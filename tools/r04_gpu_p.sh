#!/bin/bash
# round 4, call P: the whole GPU suite + smoke + the default bench line, timed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04p
mkdir -p $OUT
s=$(date +%s); timeout 3400 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; e=$(date +%s); echo "gpu suite: $((e-s)) s"; tail -4 $OUT/gpu_suite.log
s=$(date +%s); python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; e=$(date +%s); echo "smoke: $((e-s)) s rc=$?"; tail -3 $OUT/smoke.log
s=$(date +%s); python bench.py > $OUT/bench.json 2> $OUT/bench.stderr.log; e=$(date +%s); echo "bench: $((e-s)) s"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:40])
for k,v in d["secondary"].items():
    if isinstance(v,dict) and "value" in v: print(k, "%.4g" % v["value"], "ms %.3f" % v["ms_per_step"], (v.get("roofline") or {}).get("frac"), v.get("rmse_minus_sequential"), v.get("windows_per_pass"))
    else: print(k, str(v)[:300])
PY
