#!/bin/bash
# Dev A/B of the dense factorisation: bench_dense at D = 2313 and 9216 (factor ms, residual) + the stamps of a fused step.
# usage: scripts/dense_ab.sh <tag>
R=$PWD; O=$R/gpurun_out/${1:-dab}; mkdir -p $O
for D in 2313 2313 9216; do timeout -k 10 120 $R/scripts/bench_dense.bin $D 2>&1 | grep -E "^fused|^relative" | sed "s/^/D=$D /" | tee -a $O/dense.log || exit 1; done
timeout -k 10 120 $R/scripts/bench_dense_stamp.bin 2313 2>&1 | grep -E "fused step p=8" | tee -a $O/stamps.log
