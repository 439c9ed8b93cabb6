cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05c
( time timeout 1500 python bench.py --pmc off > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.stderr.log ) 2> gpurun_out/r05c/time.txt
grep "^\[bench\]\|svdfeature_amd\]" gpurun_out/r05c/bench.stderr.log | cut -c1-250 | grep -v "ranker\|evaluate\|model init" | tail -40
tail -3 gpurun_out/r05c/bench.stderr.log | cut -c1-300
cat gpurun_out/r05c/time.txt
