#!/bin/bash
# SQ counters per kernel for a bench workload (two rocprofv3 --pmc passes); prints the ratios that say what waves wait for
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/sq; rm -rf $out; mkdir -p $out
args=("$@")
pass() { name=$1; ctr=$2; d=$out/$name; mkdir -p $d; cd /tmp; export TMPDIR=/tmp; cd $root
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python3 bench.py "${args[@]}" --steps 8 --warmup 30 \
      --no-cpu-baseline --no-latency --profile-steps 0 > $d/bench.json 2> $d/err.txt || echo "pass $name FAILED rc=$?"; }
pass sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pass sq_b "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
python3 - <<PY
import csv, glob, collections
k = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES",): n[r["Kernel_Name"]] += 1
for name, c in sorted(k.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("%-34s launches %4d  waves/launch %7.0f  wave-cycles/launch %.3g | of wave cycles: wait_any %.2f wait_inst %.2f active %.2f valu %.2f | VALU insts/wave %.0f SALU %.0f LDS %.0f VMEM rd %.1f wr %.1f | lds conflict/active %.2f wait_inst_lds %.2f" % (
        name[:34], n[name], c["SQ_WAVES"] / max(1, n[name]), wc / max(1, n[name]), c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc,
        c["SQ_INSTS_VALU"] / max(1, c["SQ_WAVES"]), c["SQ_INSTS_SALU"] / max(1, c["SQ_WAVES"]), c["SQ_INSTS_LDS"] / max(1, c["SQ_WAVES"]), c["SQ_INSTS_VMEM_RD"] / max(1, c["SQ_WAVES"]), c["SQ_INSTS_VMEM_WR"] / max(1, c["SQ_WAVES"]),
        c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"]), c["SQ_WAIT_INST_LDS"] / wc))
PY
find $out -name '*.csv' -size +30M -delete
exit 0
