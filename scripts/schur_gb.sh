#!/bin/bash
# Dev A/B: batch depth of k_schur_pairs (BA_SCHUR_GB) x workgroups per CU (BA_SCHUR_WGS) at configs 5 and 4.  usage: scripts/schur_gb.sh <tag>
R=$PWD; O=$R/gpurun_out/${1:-gb}; mkdir -p $O
run() { # cfg N M K seed kind wgs gb
  BA_SCHUR_WGS=$7 BA_SCHUR_GB=$8 timeout -k 10 300 python3 - <<PY | tee -a $O/gb.log
import sys; sys.path.insert(0, "$R")
import bundleadjustment_benchmarks_amd as ba
p = ba.Problem.synthetic($2, $3, $4, $5)
s = ba.Solver(p, ba.$6, ba.F64)
s.linearize(True); s.try_step(1e-4)
print("$1 wgs/CU $7 GB $8: schur_assembly %.4f ms" % s.time_phase(3, 20, 1e-4))
PY
}
for wgs in 2 3; do for gb in 2 4 8; do run cfg5 1024 500000 4000000 1005 QRCHOL $wgs $gb; done; done
for wgs in 2 3 4; do for gb in 2 4; do run cfg4 257 65132 225911 1004 CHOLESKY $wgs $gb; done; done
run cfg4 257 65132 225911 1004 CHOLESKY 1 8
