#!/bin/bash
# Dev tool: per-launch durations of the dense factorisation at panel-bound sizes (k_ldlt_step), rocprofv3 kernel trace of bench_dense.bin.
# usage: scripts/dense_trace1.sh <tag> <binary> <D>   -> gpurun_out/<tag>/trace.csv + a per-step table
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- $R/scripts/$2 $3 > $O/run.log 2>&1
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY' | tee $O/steps.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_ldlt_step" in r["Kernel_Name"]]
# the runs of consecutive step launches = factorisations; take the 4th (warm)
runs = []
for i in idx:
    if runs and i == runs[-1][-1] + 1: runs[-1].append(i)
    else: runs.append([i])
runs = [r for r in runs if len(r) > 8]
seq = [rows[i] for i in runs[min(3, len(runs) - 1)]]
first = rows[runs[min(3, len(runs) - 1)][0] - 1]
print("panel launch before:", first["Kernel_Name"][:40], (int(first["End_Timestamp"]) - int(first["Start_Timestamp"])) / 1e3, "us")
t0 = int(first["Start_Timestamp"])
print("launches", len(seq) + 1, "span %.3f ms" % ((int(seq[-1]["End_Timestamp"]) - t0) / 1e6))
tot = gaps = 0
for k, r in enumerate(seq):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = (int(r["Start_Timestamp"]) - int((seq[k - 1] if k else first)["End_Timestamp"])) / 1e3
    tot += d; gaps += g
    print("p=%3d  grid %6s wgs  dur %6.2f us  gap %5.2f us" % (k + 1, int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) // 256, d, g))
print("sum of durations %.1f us, sum of gaps %.1f us" % (tot, gaps))
PY
