#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_init.py tests/test_gpu_dropin_cli.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -15 | tee gpurun_out/load_tests.log
python - <<'P' 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/load_probe.txt
import sys, time, json, os
sys.path.insert(0, ".")
import svdfeature_amd as sa
t = sa.Trainer(0, 0); t.seed(10)
for k, v in dict(num_user=1000000, num_item=100000, num_global=0, num_factor=64).items(): t.set_param(k, str(v))
t.init_model(); t.init_trainer()
p = "/dev/shm/probe.model"
t0 = time.perf_counter(); t.save_model(p); s = time.perf_counter() - t0
res = {"model_bytes": os.path.getsize(p), "save_model_s": round(s, 4)}
for dev in (0, 1, 1):
    b = sa.Trainer(0, 0); b.set_knob("device_load", dev)
    t0 = time.perf_counter(); b.load_model(p); b.init_trainer(); b.synchronize(); d = time.perf_counter() - t0
    res["load_model+init_trainer_s (device_load=%d)" % dev] = round(d, 4)
os.remove(p)
print(json.dumps(res))
P
