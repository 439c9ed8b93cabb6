cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_round.sh suite r05_mid
