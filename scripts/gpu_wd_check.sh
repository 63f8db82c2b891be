#!/bin/bash
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.3e pairs/s  %s' % (d['ms_per_step'], d['value'], d['config']['path']))"; }
for cfg in "--batches 2048 --overlap 1" "--batches 4096 --overlap 1" "--batches 16384 --overlap 1"; do
  for wd in 16 32; do
    echo "$cfg wd $wd: $(EGGSIM_LEVELS_WD=$wd python bench.py $cfg --steps 60 --warmup 10 --no-latency --no-cpu-baseline --profile-steps 0 2>/dev/null | ms)"
  done
done
