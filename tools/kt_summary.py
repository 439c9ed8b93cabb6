import csv, glob, statistics as st, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
g = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
g = [x for x in g if x < 1e5]
print(sys.argv[2], "dispatches", len(d), "duration us mean %.2f median %.2f" % (st.mean(d) / 1e3, st.median(d) / 1e3), "gap us mean %.2f median %.2f" % (st.mean(g) / 1e3, st.median(g) / 1e3))
