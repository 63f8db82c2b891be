"""Sums rocprofv3 --pmc counters per kernel name: python scripts/summarize_pmc.py <dir with *counter_collection.csv>"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
print("kernel,calls," + ",".join(names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], 0))):
    print(k[:40] + "," + str(len(calls[k])) + "," + ",".join("%.4g" % acc[k].get(c, 0) for c in names))
