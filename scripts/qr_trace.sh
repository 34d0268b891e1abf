#!/bin/bash
# Dev tool: per-launch timeline of QRKIT's dense QR inside one LM trial (rocprofv3 kernel trace of bench.py --workload cfg3).
# usage: scripts/qr_trace.sh <tag>   -> gpurun_out/<tag>/prof + a table of the first panels
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o cfg3 --output-format csv -- python3 $R/bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python3 - "$O/prof/cfg3_kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_qrkit_build" in r["Kernel_Name"]]
a, b = idx[5], idx[6]
seq = [r for r in rows[a:b] if "k_qr_" in r["Kernel_Name"]]
t0 = int(seq[0]["Start_Timestamp"])
for r in seq[:18] + seq[-9:]:
    print(r["Kernel_Name"][5:26], r["Grid_Size_X"], r["Workgroup_Size_X"], "start %.1f dur %.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("span ms", (int(seq[-1]["End_Timestamp"]) - t0) / 1e6, "launches", len(seq))
PY
