cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "phase_ms", {k: round(v,3) for k,v in p.items() if k!="what"})'
common="--steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --pmc off --secondary \"\" --no-window-step --exchange stratified --chunks 64 --stratified-per-item 32 --ratings 12500000 --users 125000 --items 6250"
for c in fp32 bf16; do
  eval python bench.py $common --contrib $c 2>/dev/null | python -c "$show" "rank-of-8 stratified contrib $c"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04o/kt -o kt -- python bench.py $(eval echo $common) --contrib bf16 > /dev/null 2>&1
find gpurun_out/r04o/kt -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-230
rm -rf gpurun_out/r04o/kt
