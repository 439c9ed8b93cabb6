#!/bin/bash
# round 4: window sequences of user units built on several host threads -- tests, fuzz, then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_wunit.py tests/test_gpu_native_multi.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|FAILED|assert" | tail -15 | tee gpurun_out/wseq_tests.log
FUZZ_WUNIT_DEBUG=1 timeout 900 python tests/fuzz_wunit.py --one-gpu --iters 150 --seed 91 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/wseq_fuzz.log
timeout 1200 python bench.py > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.stderr.log
tail -c 600 gpurun_out/bench_r04b.stderr.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_r04b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('init_model'), d['parity'])
for k,v in d['secondary'].items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('build_s'), v.get('schedule_build_s'), (v.get('parity') or {}).get('bit_exact'), v.get('rmse_minus_sequential'))
P
