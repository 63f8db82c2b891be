#!/bin/bash
# kernel timeline of one bench.py step (rocprofv3 --kernel-trace); arguments go to bench.py
root=${GRAFT_REPO_ROOT:-$PWD}
d=$root/gpurun_out/prof_step
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 bench.py "$@" --steps 12 --warmup 30 --no-cpu-baseline --no-latency --profile-steps 0 > $d/bench.json 2> $d/err.txt || echo "FAILED rc=$?"
f=$(find $d -name '*kernel_trace.csv' | head -1)
python3 scripts/trace_step.py $f -3 > $d/step.txt
python3 scripts/trace_gaps.py $f > $d/gaps.txt
find $d -name '*kernel_trace.csv' -size +20M -delete
exit 0
