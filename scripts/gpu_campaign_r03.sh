#!/bin/bash
# the round-3 measurement campaign in one GPU call: sweep, rocprofv3 kernel stats of the default run and of 16,384 batches,
# counter passes of the three workloads bench.py attaches counters to, the microbenchmarks, the default bench line
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
bash scripts/gpu_sweep_r03.sh > gpurun_out/campaign_sweep.log 2>&1
echo "sweep done"
bash scripts/gpu_prof.sh cfg3 > gpurun_out/campaign_prof_cfg3.log 2>&1
bash scripts/gpu_prof.sh b16k --batches 16384 --overlap 1 > gpurun_out/campaign_prof_b16k.log 2>&1
echo "profiles done"
bash scripts/collect_counters.sh cfg3 --batches 4096 --overlap 4 > gpurun_out/campaign_cnt_cfg3.log 2>&1
bash scripts/collect_counters.sh b16k --batches 16384 --overlap 1 > gpurun_out/campaign_cnt_b16k.log 2>&1
bash scripts/collect_counters.sh b4k --batches 4096 --overlap 1 > gpurun_out/campaign_cnt_b4k.log 2>&1
echo "counters done"
timeout -k 10 120 scripts/micro/build/chain_floor > gpurun_out/chain_floor.txt 2>&1
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.json | cut -c1-300
exit 0
