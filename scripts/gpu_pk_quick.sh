#!/bin/bash
# quick A/B of the packed pipeline: gpu_pk_quick.sh TAG "<bench args>" ...   (each argument one configuration)
tag=$1; shift
out=gpurun_out/pk_quick_$tag.txt
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  timeout -k 10 200 python bench.py $cfg --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 >> $out || echo "FAILED rc=$?" >> $out
done
python - $out <<'PY'
import json,sys
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith("=="): print(line, end="  ")
    elif line.startswith("{"):
        d=json.loads(line); print("ms/step %.3f  pairs/s %.3e  kernel_ms %.3f redo %d" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["config"]["redo_steps"]))
    elif line: print(line)
PY
