#!/bin/bash
# rocprofv3 counter passes for one bench workload (run on the GPU box, from the repo root):
#   scripts/collect_counters.sh TAG <bench args...>        e.g.  scripts/collect_counters.sh cfg3 --batches 4096 --overlap 4
# One pass per counter set (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ has 8 slots), no trace flags besides
# --kernel-trace, the program directly behind `--`.  Raw per-dispatch CSVs land in gpurun_out/cnt_TAG/<set>/,
# scripts/counters_to_json.py turns them into the entry of profiles/r03_counters.json that bench.py attaches.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/cnt_$tag
rm -rf $out; mkdir -p $out
steps=12; warm=3
run() {  # name, counters
  d=$out/$1; mkdir -p $d
  cd /tmp && export TMPDIR=/tmp
  cd $root
  timeout -k 10 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $d -o p -- python3 bench.py "$@" --steps $steps --warmup $warm \
      --no-cpu-baseline --no-latency --no-windows --profile-steps 0 > $d/bench.json 2> $d/err.txt || echo "pass $1 FAILED rc=$?"
}
args=("$@")
pass() { name=$1; ctr=$2; d=$out/$name; mkdir -p $d; cd /tmp; export TMPDIR=/tmp; cd $root
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python3 bench.py "${args[@]}" --steps $steps --warmup $warm \
      --no-cpu-baseline --no-latency --no-windows --profile-steps 0 > $d/bench.json 2> $d/err.txt || echo "pass $name FAILED rc=$?"; }
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
pass sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pass sq_b "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
python3 scripts/counters_to_json.py $out $tag $((steps + warm)) "${args[@]}" > $out/summary.json
cat $out/summary.json | head -60
find $out -name '*.csv' -size +40M -delete
exit 0
