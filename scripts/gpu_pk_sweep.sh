#!/bin/bash
# packed pipeline A/B: python bench.py at several sizes with the packed pipeline off / on
set -o pipefail
out=gpurun_out/pk_sweep_$1.txt
: > $out
for cfg in "--batches 1024" "--batches 4096" "--batches 16384" "--batches 4096 --overlap 4"; do
  for pk in 0 1; do
    echo "== $cfg --packed $pk" >> $out
    timeout -k 10 200 python bench.py $cfg --packed $pk --steps 40 --warmup 8 --no-cpu-baseline >> $out 2>&1 || echo "FAILED rc=$?" >> $out
  done
done
python - <<'PY' $out
import json,sys
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith("=="): print(line, end="  ")
    elif line.startswith("{"):
        d=json.loads(line); print("ms/step %.3f  pairs/s %.3e  kernel_ms %.3f redo %d" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["config"]["redo_steps"]))
    elif line: print(line)
PY
