"""Developer: the figures of a bench.py JSON line that matter at a glance."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g %s | ms/step %.3f (warm-up %d, steps %d) | frac %.4f (kernel_ms %.3f) | retiles %s redo %s levels %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["warmup"], d["steps"], r["frac"], r["kernel_ms"], d["config"].get("retiles"), d["config"].get("redo_steps"), d["config"].get("levels_per_pass")))
print("kernel:", r["kernel"])
if r.get("dominant_kernel"):
    k = r["dominant_kernel"]
    print("dominant: %s avg %.3f ms x %.0f per step, share %.2f" % (k["kernel"], k["avg_launch_ms"], k["launches_per_step"], k["share_of_launch_time"]))
for key in ("config3_protocol", "config3_late"):
    if key in d:
        w = d[key]
        print(key, w if isinstance(w, str) else "ms/step %.3f frac %.4f levels %s (warm-up %d, steps %d)" % (w["ms_per_step"], w["frac"], w["levels_per_pass"], w["warmup"], w["steps"]))
if d.get("latency_config2"):
    l = d["latency_config2"]
    print("config 2: %.3f ms/step, %.0f steps/s, kernel %.3f ms, frac %.4f" % (l["ms_per_step"], l["steps_per_sec"], l["kernel_ms"], l["roofline_frac"]))
if d.get("cpu_baseline"):
    c = d["cpu_baseline"]
    print("cpu: %.3g %s, %.0f batch-steps/s on %d core" % (c["value"], c["unit"], c["batch_steps_per_sec"], c["cores"]))
print("traffic", r.get("traffic"), (r.get("traffic_source") or {}).get("file"), (r.get("traffic_source") or {}).get("commit"))
