#!/bin/bash
# developer: fused vs packed on dense islands (four coincident blobs per site) around the automatic switch
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.3e pairs/s  %s tiles %s' % (d['ms_per_step'], d['value'], d['config']['path'], d['config']['tiles']))"; }
for b in 768 1024 1280; do
  for pk in 0 1; do
    echo "--batches $b --overlap 4 packed $pk: $(python bench.py --batches $b --overlap 4 --packed $pk --steps 60 --warmup 10 --no-latency --no-cpu-baseline --profile-steps 0 2>/dev/null | ms)"
  done
done
