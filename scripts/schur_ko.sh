#!/bin/bash
# Dev experiment: the Schur assembly at config 5 with one side's record gathers knocked out (always cache hits): what a row-major /
# column-stationary order could save AT MOST.  usage: scripts/schur_ko.sh <tag>
R=$PWD; O=$R/gpurun_out/${1:-ko}; mkdir -p $O
for ko in 0 1 2 3; do
  BA_SCHUR_KNOCKOUT=$ko timeout -k 10 300 python3 - <<PY | tee -a $O/ko.log
import sys; sys.path.insert(0, "$R")
import bundleadjustment_benchmarks_amd as ba
p = ba.Problem.synthetic(1024, 500000, 4000000, 1005)
s = ba.Solver(p, ba.QRCHOL, ba.F64)
s.linearize(True); s.try_step(1e-4)
print("knockout $ko: schur_assembly %.3f ms" % s.time_phase(3, 10, 1e-4))
PY
done
