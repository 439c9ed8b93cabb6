#!/bin/bash
# Run on the GPU box through gpurun:  tools/profile_round.sh r01
# Produces gpurun_out/<tag>/ : bench JSON, rocprofv3 kernel stats, PMC summaries, hbm_traffic.json.
# Counters are collected in their own passes (never together with sys/hip/hsa tracing).
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.stderr.log
tail -1 $OUT/bench.json | cut -c1-300
# per-kernel time of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt.stderr.log
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats.csv
python - <<PY > $OUT/kernel_trace_summary.txt
import csv, glob, statistics as st
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_basicmf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
g = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
g = [x for x in g if x < 1e6]
print("kernel k_basicmf dispatches", len(d))
print("duration us: mean %.3f median %.3f min %.3f max %.3f" % (st.mean(d) / 1e3, st.median(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
print("gap to next dispatch us: mean %.3f median %.3f" % (st.mean(g) / 1e3, st.median(g) / 1e3))
r = rows[len(rows) // 2]
print("launch shape:", {k: r[k] for k in ("Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")})
PY
cat $OUT/kernel_trace_summary.txt
rm -rf $OUT/kt
# hardware counters: one pass per counter group, 1 pass over the data, no warm-up
: > $OUT/pmc_summary.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  n=$(echo $c | tr " " "_")
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_$n.stderr.log
  python tools/pmc_summary.py $OUT/pmc_$n | grep -v copyBuffer >> $OUT/pmc_summary.txt
  rm -rf $OUT/pmc_$n $OUT/pmc_$n.stderr.log
done
cat $OUT/pmc_summary.txt
python - <<PY
import json, re
txt = open("$OUT/pmc_summary.txt").read()
def mean(counter):
    m = re.search(r"k_basicmf[^\n]*?\s%s\s+n=\s*\d+\s+mean=(\S+)" % counter, txt)
    return float(m.group(1))
fetch_kb, write_kb = mean("FETCH_SIZE"), mean("WRITE_SIZE")
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes
# of wide (16 B/lane) coalesced reads -> doubled.  WRITE_SIZE is used as reported.
out = {"kernel": "k_basicmf", "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
       "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
       "tcc_ea_rdreq_per_launch": mean("TCC_EA0_RDREQ_sum"), "tcc_ea_wrreq_per_launch": mean("TCC_EA0_WRREQ_sum"),
       "tcc_hit_per_launch": mean("TCC_HIT_sum"), "tcc_miss_per_launch": mean("TCC_MISS_sum"),
       "note": "rocprofv3 --pmc, separate passes, bench.py --steps 1 --warmup 0, mean over all k_basicmf dispatches; "
               "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction"}
json.dump(out, open("$OUT/hbm_traffic.json", "w"), indent=1)
print(out)
PY
