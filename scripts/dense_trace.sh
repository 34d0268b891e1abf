#!/bin/bash
# Dev tool: per-launch durations of the dense factorisation (rocprofv3 kernel trace of scripts/bench_dense*.bin).
# usage: scripts/dense_trace.sh <tag> <binary> <D>   -> gpurun_out/<tag>/trace.csv + a per-step table
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- $R/scripts/$2 $3 > $O/run.log 2>&1
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete fused factorisation: the last run of consecutive k_ldlt_step2 launches
idx = [i for i, r in enumerate(rows) if "k_ldlt_step2" in r["Kernel_Name"]]
if not idx:
    print("no k_ldlt_step2 launches"); sys.exit(0)
end = idx[-1]
start = end
while start - 1 >= 0 and "k_ldlt_step" in rows[start - 1]["Kernel_Name"]: start -= 1
seq = rows[start:end + 1]
t0 = int(seq[0]["Start_Timestamp"])
print("launches", len(seq), "span %.3f ms" % ((int(seq[-1]["End_Timestamp"]) - t0) / 1e6))
for k, r in enumerate(seq):
    if k < 12 or k % 8 < 2 or k > len(seq) - 6:
        print("p=%3d  grid %6s  dur %8.1f us  gap %5.1f us" % (k + 1, r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
              (int(r["Start_Timestamp"]) - int(seq[k - 1]["End_Timestamp"])) / 1e3 if k else 0.0))
PY
