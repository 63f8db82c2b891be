"""Developer: from a rocprofv3 kernel trace (csv), the idle time between consecutive kernels of each queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in byq.items():
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]
    print("queue %s: %d kernels, busy %.3f ms, gaps < 50 us: %d, mean %.2f us, median %.2f us, total %.3f ms" % (
        q, len(ks), busy / 1e6, len(small), sum(small) / max(1, len(small)) / 1e3, sorted(small)[len(small) // 2] / 1e3 if small else 0, sum(small) / 1e6))
    by = collections.defaultdict(list)
    for i, g in enumerate(gaps):
        if 0 <= g < 50000: by[ks[i + 1][2][:40]].append(g)
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print("   before %-42s n=%5d mean %.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
