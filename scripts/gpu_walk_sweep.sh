#!/bin/bash
# in-order against out-of-order level walk on separate blobs (one process per size, variants interleaved)
for nb in 2048 4096 8192 16384; do
  echo "== $nb separate batches"
  timeout -k 10 200 python scripts/gpu_ab.py --batches $nb --overlap 1 --warmup 10 --steps 45 --rounds 3 --variants walk=1 walk=2 2>&1 | grep -E "white stream|per kind|bit-" | sed "s/yolk:.*//"
done
