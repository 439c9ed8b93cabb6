#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE, one pass each) for the f3 kernels: k_predict_basic (evaluator, 20 M instances) and the ranker's scoring
# kernels (100 K candidates, k = 128: tiled and untiled).  -> gpurun_out/<tag>/pmc_f3.txt and f3_traffic.json (merged into profiles/hbm_traffic.json)
set -u
TAG=${1:-r03pmc}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
F=$OUT/pmc_f3.txt
: > $F
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmce_$c -o p -- python tests/perf_eval.py --ratings 20000000 --cpu-sample 1000 --reps 2 > /dev/null 2> $OUT/pmce.stderr.log
  python tools/pmc_summary.py $OUT/pmce_$c | grep -E "k_predict_basic|counter_collection" >> $F
  rm -rf $OUT/pmce_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcr_$c -o p -- python tests/perf_ranker.py --sections 200 --cpu-sections 1 > /dev/null 2> $OUT/pmcr.stderr.log
  python tools/pmc_summary.py $OUT/pmcr_$c | grep -E "k_rank_score|counter_collection" >> $F
  rm -rf $OUT/pmcr_$c
done
cat $F
python - <<PY
import re, json
txt = open("$F").read()
def mean(kern, counter):
    tot, n = 0.0, 0
    for m in re.finditer(r"%s[^\n]*?\s%s\s+n=\s*(\d+)\s+mean=(\S+)\s+total=(\S+)" % (kern, counter), txt):
        n += int(m.group(1)); tot += float(m.group(3))
    return tot / n if n else None
out = {}
for key, kern in (("evaluate_k64", "k_predict_basic"), ("ranker_k128_positions_tile", "k_rank_score_tile"), ("ranker_k128_positions_single", "k_rank_score<8, 1>")):
    f, w = mean(re.escape(kern), "FETCH_SIZE"), mean(re.escape(kern), "WRITE_SIZE")
    if f is not None and w is not None:
        out[key] = {"kernel": kern, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(out, open("$OUT/f3_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
