#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svdf::", "")[-56:], r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    print(f)
    for (kn, cn), (n, tot) in sorted(acc.items()):
        print("  %-42s %-22s n=%6d mean=%.4g total=%.6g" % (kn, cn, n, tot / n, tot))
