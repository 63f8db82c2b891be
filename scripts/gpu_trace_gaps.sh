#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
d=$root/gpurun_out/prof_gaps
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --no-latency --profile-steps 0 > $d/bench.json 2> $d/err.txt || echo "FAILED rc=$?"
f=$(find $d -name '*kernel_trace.csv' | head -1)
head -1 $f | tr ',' '\n' | head -30 > $d/cols.txt
python3 scripts/trace_gaps.py $f
find $d -name '*kernel_trace.csv' -size +20M -delete
exit 0
