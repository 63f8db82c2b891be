#!/bin/bash
# rocprofv3 kernel trace of a short bench run; usage: gpu_prof.sh TAG <bench args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
d=$root/gpurun_out/prof_$tag
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py "$@" --no-cpu-baseline > $d/bench.json 2> $d/err.txt || echo "FAILED rc=$?"
f=$(find $d -name '*kernel_stats.csv' | head -1)
echo "== $tag: $@"; tail -1 $d/bench.json | cut -c1-200
if [ -n "$f" ]; then cut -d, -f1-7 "$f" | head -16; cp "$f" $d/kernel_stats.csv; fi
t=$(find $d -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then python3 scripts/summarize_profile.py "$t" > $d/by_launch.md; fi
find $d -name '*kernel_trace.csv' -size +20M -delete
exit 0
