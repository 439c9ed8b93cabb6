#!/bin/bash
# round 4, call N: k_fewrow_gslots parity + A/B against k_fused on the neighbourhood workload (level time, counters), the fixed drop-in test
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04n
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sched.py tests/test_gpu_dropin_cli.py -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d["roofline"]; print(sys.argv[1], "M inst/s %.2f" % (d["value"]/1e6), "ms/pass %.2f" % d["ms_per_step"], "us/level %.3f" % r["avg_launch_us"], "levels", d["config"]["conflict_free_batches_per_pass"], "parity", d.get("parity"), "dag", (d.get("dag_bound") or {}).get("unit_latency_us"))'
for kn in 1 0; do
  timeout 600 python bench.py --workload neighbourhood --secondary "" --pmc off --steps 5 --knob fewrow_gslots=$kn 2> $OUT/neigh_$kn.log | python -c "$show" "neighbourhood exact, fewrow_gslots=$kn" | tee -a $OUT/ab.txt
done
for kn in 1 0; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_MISSES SQC_DCACHE_MISSES"; do
    n=$(echo $c | tr " " "_")
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${kn}_$n -o p -- python bench.py --workload neighbourhood --secondary "" --pmc off --no-cpu-baseline --steps 1 --warmup 0 --knob fewrow_gslots=$kn > /dev/null 2> $OUT/pmc.stderr.log
    echo "fewrow_gslots=$kn" >> $OUT/pmc_neigh.txt
    python tools/pmc_summary.py $OUT/pmc_${kn}_$n | grep -E "k_fused|k_fewrow" >> $OUT/pmc_neigh.txt
    rm -rf $OUT/pmc_${kn}_$n
  done
done
cat $OUT/pmc_neigh.txt
