#!/bin/bash
# batch-count sweep for profiles/r03_batch_sweep.md (run on the GPU box)
out=gpurun_out/sweep_r03.txt
: > $out
for cfg in "--batches 256 --overlap 1" "--batches 1024 --overlap 1" "--batches 2048 --overlap 1" "--batches 4096 --overlap 1" "--batches 8192 --overlap 1" "--batches 16384 --overlap 1" "--batches 65536 --overlap 1" "--batches 4096 --overlap 4" "--batches 16384 --overlap 4"; do
  echo "== $cfg" >> $out
  timeout -k 10 300 python bench.py $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-latency --no-windows --profile-steps 0 2>/dev/null | tail -1 >> $out || echo "FAILED rc=$?" >> $out
done
python - $out > gpurun_out/sweep_r03.md <<'PY'
import json,sys
print("| workload | particles | path | tiles white/yolk | steps/s | pair-solves/s | ms/step | kernel ms/step (white launches) | host gap | roofline frac (592 B model) |")
print("|---|---|---|---|---|---|---|---|---|---|")
cfg=None
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith("=="): cfg=line[3:]
    elif line.startswith("{"):
        d=json.loads(line); r=d["roofline"]
        gap=(d["ms_per_step"]-r["kernel_ms_timed_region"])/d["ms_per_step"]
        print("| %s | %d | %s | %s / %s | %.0f | %.3g | %.3f | %.3f | %.1f %% | %.4f |" % (cfg, d["particles"], d["config"]["path"], d["config"]["tiles"][0], d["config"]["tiles"][1], d["steps_per_sec"], d["value"], d["ms_per_step"], r["kernel_ms_timed_region"], 100*gap, r["frac"]))
    elif line: print(cfg, line)
PY

cat gpurun_out/sweep_r03.md
