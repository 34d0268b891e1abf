#!/bin/bash
# rocprofv3 kernel stats of bench.py for one workload: scripts/gpu_prof.sh <tag> <workload> [steps]
set -o pipefail
R=$PWD
TAG=${1:-prof}; WL=${2:-cfg4}; STEPS=${3:-40}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_$WL -o $WL --output-format csv -- python3 $R/bench.py --workload $WL --steps $STEPS --warmup 4 --no-cpu-baseline --phase-reps 4 > $O/prof_$WL.log 2>&1
echo "rocprof rc=$?"
F=$(find $O/prof_$WL -name "*kernel_stats.csv" | head -1)
cp $F $O/${WL}_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/${WL}_kernel_stats.csv")))
for r in rows[:16]:
    print("%-90s calls %6s  avg %10.2f us  total%% %s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $O/prof_$WL.log | cut -c1-600
