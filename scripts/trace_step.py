"""Developer: from a rocprofv3 kernel trace (csv), the kernel sequence of ONE step per queue: start offset, duration,
gap to the previous kernel of the same queue.  A step = the kernels between two egg_pk_begin_kernel launches of the
busiest queue."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3  # which step (negative: from the end)
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), int(r.get("Grid_Size", 0) or 0)) for r in rows)
busy = collections.Counter()
for s, e, n, q, g in ks:
    busy[q] += e - s
main_q = busy.most_common(1)[0][0]
begins = [s for s, e, n, q, g in ks if q == main_q and n.startswith("egg_pk_begin")]
t0, t1 = begins[which], begins[which + 1]
print("step of %.3f ms on queue %s" % ((t1 - t0) / 1e6, main_q))
last_end = {}
for s, e, n, q, g in ks:
    if t0 <= s < t1:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        print("q%-3s +%8.1f us  dur %8.1f us  gap %6.1f  grid %8d  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, g, n[:48]))
    last_end[q] = max(last_end.get(q, 0), e) if t0 <= s else e
