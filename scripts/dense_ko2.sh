#!/bin/bash
# Dev tool: knock-outs of the macro-tile update of k_ldlt_step2 (wrong results, the time tells) at D = 9216.
# build: scripts/bench_dense.hip with -DBA_KO_MACRO_C / _STAGE / _MFMA (ba_dense.hip.h: ba_update_macro) -> scripts/bd_ko_*.bin
O=gpurun_out/${1:-ko2}; mkdir -p $O
for b in bench_dense bd_ko_c bd_ko_stage bd_ko_mfma bd_ko_c_stage; do
  echo "== $b" >> $O/out.txt
  timeout -k 10 120 scripts/$b.bin 9216 2>&1 | grep "factor" | head -2 >> $O/out.txt
done
cat $O/out.txt
