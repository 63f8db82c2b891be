"""Summarises a rocprofv3 --kernel-trace CSV per (kernel, workgroup size): the white and yolk
launches of egg_step_kernel share a name and differ in workgroup size.
    python scripts/summarize_profile.py gpurun_out/prof_bench/runc/*_kernel_trace.csv > profiles/<name>.md"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        key = (r["Kernel_Name"], int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"]),
               int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]),
               r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | workgroup | grid (threads) | LDS B/block | VGPRs | calls | avg us | min us | max us | total ms |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("| %s | %d | %d | %s | %s | %d | %.1f | %.1f | %.1f | %.2f |" %
          (key[0], key[1], key[2], key[3], key[4], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(d) / 1e6))
