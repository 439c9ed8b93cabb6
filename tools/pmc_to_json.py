#!/usr/bin/env python3
"""pmc_<workload>.txt (tools/profile_round2.sh) -> hbm_traffic.json: per-launch means of the dominant kernel's counters.
usage: python tools/pmc_to_json.py <dir with pmc_*.txt or r02_pmc_*.txt> <out.json>"""
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
names = {"basicmf": "k_basicmf", "pairwise": "k_fewrow", "svdpp": "k_svdpp_wave", "neighbourhood": "k_fused"}   # prefixes the summaries are matched by
full = {"basicmf": "svdf::k_basicmf_slots<8, 2, G> (G = 1 ... 4 row sets per wave, chosen per launch size; all instantiations pooled)",
        "pairwise": "svdf::k_fewrow_slots<16, 2, 1, 2>", "svdpp": "svdf::k_svdpp_wave<2, true, true, true, false, 8>",
        "neighbourhood": "svdf::k_fused<32, 1, 1, 1, true, false>"}   # the names in profiles/r0N_kernel_stats.csv
out = {}
for w, kern in names.items():
    cand = glob.glob(os.path.join(src, "*pmc_%s.txt" % w))
    if not cand:
        continue
    txt = open(cand[0]).read()

    def mean(counter):
        # every instantiation of the kernel (e.g. k_basicmf_slots<8, 2, G> for the G chosen per launch): total / dispatches
        tot, n = 0.0, 0
        for m in re.finditer(r"%s[^\n]*?\s%s\s+n=\s*(\d+)\s+mean=(\S+)\s+total=(\S+)" % (kern, counter), txt):
            n += int(m.group(1))
            tot += float(m.group(3))
        return (tot / n, n) if n else (None, 0)
    f, nf = mean("FETCH_SIZE")
    wr, _ = mean("WRITE_SIZE")
    if f is None or wr is None:
        continue
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes
    # of wide (16 B/lane) coalesced reads -> doubled.  WRITE_SIZE is used as reported.
    out[w] = {"kernel": full[w], "dispatches": nf, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": wr,
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
json.dump(flat, open(dst, "w"), indent=1)
print(json.dumps(flat)[:2000])
