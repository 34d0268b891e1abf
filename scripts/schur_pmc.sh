#!/bin/bash
# Dev tool: PMC passes over the Schur-assembly kernel variants at one workload.  usage: scripts/schur_pmc.sh <tag> <workload>
R=$PWD; TAG=${1:-pmc}; WL=${2:-cfg4}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/schur_one.py <<PY
import sys, os
sys.path.insert(0, "$R")
import bundleadjustment_benchmarks_amd as ba
wl = sys.argv[1]
dims = {"cfg4": (257, 65132, 225911, 1004, ba.CHOLESKY), "cfg5": (1024, 500000, 4000000, 1005, ba.QRCHOL)}[wl]
p = ba.Problem.synthetic(*dims[:4]); kind = dims[4]
s = ba.Solver(p, kind, ba.F64)
s.linearize(); s.try_step(1e-4)
print("schur %.1f us" % (1e3 * s.time_phase(3, 5, 1e-4)))
PY
for IMPL in 0 1; do
  export BA_SCHUR_BANDS=$((1 + 7 * IMPL))
  i=0
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET -d $O/i${IMPL}_s$i -o p --output-format csv -- python3 /tmp/schur_one.py $WL > $O/i${IMPL}_s$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for impl in (0, 1):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$O/i%d_s*/**/*counter_collection.csv" % impl, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_schur_pairs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("bands =", 1 + 7 * impl)
    for k in sorted(agg): print("   %-40s %14.4g per launch (%d samples)" % (k, agg[k] / max(n[k], 1), n[k]))
PY
grep -h schur $O/*.log | head -4
