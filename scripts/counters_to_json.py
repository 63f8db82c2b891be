"""Aggregates the passes of scripts/collect_counters.sh: python scripts/counters_to_json.py <dir> <tag> <steps> <bench args...>
Prints one JSON object: per-kernel counter sums and the per-step figures bench.py attaches (roofline.traffic / roofline.valu)."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
bench_args = sys.argv[4:]
per_kernel = defaultdict(lambda: defaultdict(float))
busy_ns = None
steps_per_sec = None
for sub in ("fetch", "write", "sq_a", "sq_b"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                per_kernel[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if sub == "sq_a":
        # GPU busy time of that pass: union of the dispatch intervals (two streams overlap)
        iv = []
        for f in glob.glob("%s/%s/**/*kernel_trace.csv" % (d, sub), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r["Kernel_Name"].startswith("egg_"):
                        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        iv.sort()
        busy, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        busy_ns = busy
        try:
            steps_per_sec = json.loads(open("%s/%s/bench.json" % (d, sub)).read().strip().splitlines()[-1])["steps_per_sec"]
        except Exception:
            pass
tot = defaultdict(float)
for k, c in per_kernel.items():
    if k.startswith("egg_"):
        for n, v in c.items():
            tot[n] += v
fetch_kb, write_kb = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
out = {
    "tag": tag, "bench_args": bench_args, "steps_profiled": steps,
    "method": "rocprofv3 --pmc, one pass per counter set (FETCH_SIZE | WRITE_SIZE | 8 SQ counters | 8 SQ counters), --kernel-trace only, "
              "summed over every egg_* dispatch of the run and divided by the steps run; FETCH_SIZE doubled per MI355X_MICROARCH.md "
              "(gfx950 tallies 128-B requests as 64 B; exact for wide coalesced streams, an upper estimate for the narrow reads here); "
              "Infinity-Cache hits are included, so true HBM traffic is lower",
    "hbm_bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0 / steps,
    "fetch_kb_per_step": fetch_kb / steps, "write_kb_per_step": write_kb / steps,
}
if tot.get("SQ_ACTIVE_INST_VALU"):
    valu = {
        "valu_insts_per_step": tot["SQ_INSTS_VALU"] / steps,
        "lanes_per_inst": tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"] if tot.get("SQ_THREAD_CYCLES_VALU") else None,
        # egg_pk_exec_chain_kernel (and the executor inside egg_pk_levexec_kernel) keeps all 64 lanes enabled and selects afterwards (project_pair_predicated), so its lanes
        # per instruction say nothing about useful lanes: the same ratio over every other kernel
        "lanes_useful": (lambda tc, ac: tc / ac if ac else None)(
            sum(c.get("SQ_THREAD_CYCLES_VALU", 0.0) for k, c in per_kernel.items() if k.startswith("egg_") and "exec_chain" not in k and "levexec" not in k),
            sum(c.get("SQ_ACTIVE_INST_VALU", 0.0) for k, c in per_kernel.items() if k.startswith("egg_") and "exec_chain" not in k and "levexec" not in k)),
        "lanes_useful_note": "lanes per VALU instruction over all kernels except egg_pk_exec_chain_kernel and egg_pk_levexec_kernel (their executors are predicated: every lane enabled; the walk of the latter polls with all lanes)",
        "wave_time_valu_active": tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"],
        "wave_time_any_inst_active": tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
        "wave_time_waiting": tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"],
        "wave_time_issue_stalled": tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
    }
    if busy_ns:
        # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md); 1024 SIMDs, nominal 2.4 GHz
        valu["issue_frac"] = tot["SQ_ACTIVE_INST_VALU"] * 4.0 / (busy_ns * 1e-9 * 2.4e9 * 1024)
        valu["gpu_busy_ms_per_step"] = busy_ns * 1e-6 / steps
    if steps_per_sec:
        valu["insts_per_s"] = valu["valu_insts_per_step"] * steps_per_sec
    out["valu"] = valu
out["kernels"] = {k: dict(c) for k, c in sorted(per_kernel.items()) if k.startswith("egg_")}
print(json.dumps(out, indent=1))
