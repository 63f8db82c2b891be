#!/bin/bash
# developer: executor group size sweep (particles per executor wave) on mid-size scenes
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.3e pairs/s' % (d['ms_per_step'], d['value']))"; }
for cfg in "--batches 2048 --overlap 1" "--batches 4096 --overlap 1" "--batches 8192 --overlap 1" "--batches 16384 --overlap 1" "--batches 4096 --overlap 4"; do
  for gp in 320 640 960 1280; do
    echo "$cfg gp $gp: $(python bench.py $cfg --group-particles $gp --packed 1 --steps 60 --warmup 10 --no-latency --no-cpu-baseline --profile-steps 0 2>/dev/null | ms)"
  done
done
