#!/bin/bash
# Dev A/B on ONE box: scripts/bench_dense_base.bin (the committed kernels) against scripts/bench_dense.bin (the working tree), interleaved.
R=$PWD; O=$R/gpurun_out/${1:-dab2}; mkdir -p $O
for rep in 1 2 3; do for b in base new; do
  [ $b = base ] && B=$R/scripts/bench_dense_base.bin || B=$R/scripts/bench_dense.bin
  timeout -k 10 120 $B ${2:-2313} 2>&1 | grep -E "^fused" | sed "s/^/$b /" | tee -a $O/ab.log || exit 1
done; done
