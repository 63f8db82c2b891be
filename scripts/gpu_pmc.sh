#!/bin/bash
# rocprofv3 counter collection of a short bench run (one pass per counter set); usage: gpu_pmc.sh TAG "<counters>" <bench args...>
tag=$1; shift
counters=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
d=$root/gpurun_out/pmc_$tag
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -k 10 400 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $d -o p -- python3 bench.py "$@" --no-cpu-baseline > $d/bench.json 2> $d/err.txt || echo "FAILED rc=$?"
ls $d
python3 scripts/summarize_pmc.py $d > $d/summary.txt 2>&1 || true
cat $d/summary.txt
find $d -name '*.csv' -size +30M -delete
exit 0
