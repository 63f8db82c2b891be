#!/bin/bash
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.3e pairs/s  %s' % (d['ms_per_step'], d['value'], d['config']['path']))"; }
for b in 512 1024 1536 2048 3072; do
  echo "--batches $b fused: $(python bench.py --batches $b --overlap 1 --packed 0 --steps 60 --warmup 10 --no-latency --no-cpu-baseline --profile-steps 0 2>/dev/null | ms)"
  for gp in 160 320 480; do
    echo "--batches $b packed gp $gp: $(python bench.py --batches $b --overlap 1 --group-particles $gp --packed 1 --steps 60 --warmup 10 --no-latency --no-cpu-baseline --profile-steps 0 2>/dev/null | ms)"
  done
done
