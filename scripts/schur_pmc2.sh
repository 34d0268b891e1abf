#!/bin/bash
# Dev tool: where k_schur_pairs' cycles go -- SQ and TA counters of the kernel at one workload (separate --pmc passes, kernel trace only).
# usage: scripts/schur_pmc2.sh <tag> <workload>
R=$PWD; O=$R/gpurun_out/$1; W=${2:-cfg5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W --steps 6 --warmup 1 --regions 1 --no-cpu-baseline --phase-reps 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/a -o a --output-format csv -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d $O/b -o b --output-format csv -- $B > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $O/c -o c --output-format csv -- $B > $O/c.log 2>&1
python3 - $O <<'PY' | tee $O/summary_$W.txt
import csv, glob, sys, collections
O = sys.argv[1]
for tag in "abc":
    fs = glob.glob(O + "/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_schur_pairs" not in k and "k_ldlt_step" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k[:60], {c: round(v / max(n[(k, c)], 1), 1) for c, v in acc[k].items()})
PY
rm -rf $O/a $O/b $O/c
