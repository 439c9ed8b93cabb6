cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1
tail -40 gpurun_out/r05_profile.log
