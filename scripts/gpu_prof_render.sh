#!/bin/bash
# rocprofv3 kernel trace of the renderer demo (scripts/gpu_render_demo.py): per-kernel times of draw() on config 2
root=${GRAFT_REPO_ROOT:-$PWD}
d=$root/gpurun_out/prof_render
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 scripts/gpu_render_demo.py 20 > $d/out.txt 2> $d/err.txt || echo "FAILED rc=$?"
cat $d/out.txt
f=$(find $d -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then grep -E "Name|egg_render|egg_env" "$f" | cut -d, -f1-7; cp "$f" $root/gpurun_out/render_kernel_stats.csv; fi
exit 0
