cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05h /tmp/bulk
gcc -std=gnu99 -O2 -Iinclude integration/svdf_train_bulk.c -o /tmp/bulk/svdf_train_bulk -Lsvdfeature_amd -lsvdfeature_amd -Wl,-rpath,$PWD/svdfeature_amd
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from svdfeature_amd import data as D
import svdfeature_amd as sa
u, i, r = bench.synth_triples(100_000_000, 1_000_000, 100_000, 12345)
D.write_csr_buffer("/tmp/bulk/train.buffer", sa.CSRData.from_triples(u, i, r))
open("/tmp/bulk/run.conf", "w").write("\n".join("%s = %s" % kv for kv in [("base_score", "3"), ("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", "100000"), ("num_user", "1000000"), ("num_global", "0"), ("num_factor", "64"), ("active_type", "0"), ("buffer_feature", '"train.buffer"'), ("model_out_folder", '"./"')]) + "\n")
PY
cd /tmp/bulk
for mode in async sync async sync async sync; do
  rm -f /tmp/bulk/0*.model; sync
  if [ $mode = sync ]; then export SVDF_BULK_SYNC_SAVE=1; else unset SVDF_BULK_SYNC_SAVE; fi
  echo "== $mode"; ./svdf_train_bulk run.conf num_round=8 2>&1 | grep "seconds per round"
done | tee "${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/r05h/bulk_save_ab.txt"
