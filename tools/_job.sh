cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05c
( time timeout 1500 python bench.py > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.stderr.log ) 2> gpurun_out/r05c/time.txt
grep "^\[bench\] orders\|PMC\|svdpp" gpurun_out/r05c/bench.stderr.log | cut -c1-220 | tail -14
cat gpurun_out/r05c/time.txt
