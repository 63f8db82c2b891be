"""Merges gpurun_out/cnt_*/summary.json (scripts/collect_counters.sh) into profiles/r03_counters.json, keyed
"<batches>_<overlap>" -- the file bench.py reads roofline.traffic / roofline.valu from."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "r03_counters.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "cnt_*", "summary.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print("skip", f, e)
        continue
    a = d["bench_args"]
    batches = int(a[a.index("--batches") + 1]) if "--batches" in a else 4096
    overlap = int(a[a.index("--overlap") + 1]) if "--overlap" in a else 4
    out["%d_%d" % (batches, overlap)] = d
    print("merged", f, "->", "%d_%d" % (batches, overlap))
json.dump(out, open(dst, "w"), indent=1)
