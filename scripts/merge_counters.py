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
import subprocess
try:  # the commit the counters were taken at (the GPU box has no .git: stamped here, when the summaries are merged)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "egg_fluid_simulation_amd"], capture_output=True, text=True).stdout.strip())
    out["_source"] = {"code_commit": head + ("+uncommitted changes" if dirty else ""), "how": "scripts/collect_counters.sh on one MI355X, merged by scripts/merge_counters.py"}
except Exception:
    pass
json.dump(out, open(dst, "w"), indent=1)
