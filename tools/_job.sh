cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_round.sh fuzz r05_fuzz 7000
