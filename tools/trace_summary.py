#!/usr/bin/env python3
"""Per-kernel duration / gap summary from a rocprofv3 kernel_trace.csv directory."""
import csv, glob, statistics as st, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][-48:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in by.items():
    print("%-50s n=%6d mean %.2f us median %.2f min %.2f max %.2f" % (k, len(d), st.mean(d) / 1e3, st.median(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
g = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
g = [x for x in g if x < 1e6]
print("gap between consecutive dispatches: mean %.2f us median %.2f p90 %.2f" % (st.mean(g) / 1e3, st.median(g) / 1e3, sorted(g)[int(len(g) * 0.9)] / 1e3))
