cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05f
SVDF_QUIET=1 timeout 2400 python tools/contract_seeds.py 0 2,8 --zipf 0.7 --per-item 24 --checks 3,10 2> gpurun_out/r05f/contract_zipf.err | tee gpurun_out/r05f/contract_zipf.txt
tail -3 gpurun_out/r05f/contract_zipf.err
