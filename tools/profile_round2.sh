#!/bin/bash
# Run on the GPU box through gpurun:  tools/profile_round2.sh r02
# Produces gpurun_out/<tag>/ : the bench line, rocprofv3 kernel stats of the same command (main line + secondary workloads),
# and per-workload PMC summaries (FETCH/WRITE sizes, TCC EA requests, SQ instruction counts) of its dominant kernel.
# Counters are collected in their own passes (--kernel-trace + --pmc only, never with sys/hip/hsa tracing).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.stderr.log
tail -1 $OUT/bench.json | cut -c1-300
# per-kernel time of the same command (all workloads of the default line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt.stderr.log
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
head -12 $OUT/kernel_stats.csv
# hardware counters: one pass per counter group and workload, 1 pass over the data, no warm-up
for W in basicmf pairwise svdpp neighbourhood; do
  : > $OUT/pmc_$W.txt
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    n=$(echo $c | tr " " "_")
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$n -o p -- python bench.py --workload $W --no-cpu-baseline --secondary "" --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_${W}_$n.stderr.log
    python tools/pmc_summary.py $OUT/pmc_${W}_$n | grep -E "svdf::k_|counter_collection" >> $OUT/pmc_$W.txt
    rm -rf $OUT/pmc_${W}_$n $OUT/pmc_${W}_$n.stderr.log
  done
  echo "== $W"; cat $OUT/pmc_$W.txt
done
python - <<PY
import json, re
out = {}
names = {"basicmf": "k_basicmf", "pairwise": "k_fused", "svdpp": "k_svdpp_wave", "neighbourhood": "k_fused"}
for w, kern in names.items():
    txt = open("$OUT/pmc_%s.txt" % w).read()
    def mean(counter):
        m = re.search(r"svdf::%s[^\n]*?\s%s\s+n=\s*(\d+)\s+mean=(\S+)" % (kern, counter), txt)
        return (float(m.group(2)), int(m.group(1))) if m else (None, 0)
    f, nf = mean("FETCH_SIZE"); wr, _ = mean("WRITE_SIZE")
    if f is None or wr is None:
        continue
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes
    # of wide (16 B/lane) coalesced reads -> doubled.  WRITE_SIZE is used as reported.
    out[w] = {"kernel": kern, "dispatches": nf, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": wr,
              "hbm_bytes_per_launch": (2 * f + wr) * 1024,
              "tcc_ea_rdreq_per_launch": mean("TCC_EA0_RDREQ_sum")[0], "tcc_ea_wrreq_per_launch": mean("TCC_EA0_WRREQ_sum")[0],
              "tcc_hit_per_launch": mean("TCC_HIT_sum")[0], "tcc_miss_per_launch": mean("TCC_MISS_sum")[0],
              "sq_waves_per_launch": mean("SQ_WAVES")[0], "sq_insts_valu_per_launch": mean("SQ_INSTS_VALU")[0],
              "sq_insts_salu_per_launch": mean("SQ_INSTS_SALU")[0], "sq_insts_vmem_rd_per_launch": mean("SQ_INSTS_VMEM_RD")[0],
              "sq_insts_vmem_wr_per_launch": mean("SQ_INSTS_VMEM_WR")[0]}
flat = dict(out.get("basicmf", {}))
flat.update({k: v for k, v in out.items() if k != "basicmf"})
flat["note"] = ("rocprofv3 --pmc, separate passes per counter group, bench.py --workload W --steps 1 --warmup 0, mean over all dispatches of the "
                "workload's dominant kernel; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction; top-level keys = basicMF "
                "(the contract line), nested objects = the secondary workloads")
json.dump(flat, open("$OUT/hbm_traffic.json", "w"), indent=1)
print(json.dumps(flat)[:1500])
PY
