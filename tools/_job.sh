#!/bin/bash
mkdir -p gpurun_out/r06_contract
for cfg in "24 16" "24 12"; do
  set -- $cfg
  timeout 1100 python tools/contract_seeds.py 0,1,2 8 --zipf 0.7 --skip-allreduce --checks 3 --per-item $1 --chunks $2 \
      > gpurun_out/r06_contract/zipf_c2_per${1}_chunks${2}.txt 2> gpurun_out/r06_contract/zipf_c2_per${1}_chunks${2}.err
  echo "cfg $cfg rc=$?"
  cat gpurun_out/r06_contract/zipf_c2_per${1}_chunks${2}.txt
done
