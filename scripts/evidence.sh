#!/bin/bash
# Round-2 evidence, one gpurun call: bench lines of every config, rocprofv3 kernel stats, PMC passes (HBM bytes: FETCH_SIZE and
# WRITE_SIZE in passes of their own; matrix-core busy cycles), the configs run to the reference's own stop next to the oracle.
# usage: scripts/evidence.sh <tag>   -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
R=$PWD
O=$R/gpurun_out/${1:-r02ev}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 100 --warmup 10 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
for c in cfg1 cfg2 cfg3 cfg5; do python3 $R/bench.py --workload $c --steps 100 --warmup 10 > $O/bench_$c.json 2> $O/bench_$c.err; done
echo benches done
for c in cfg4 cfg5 cfg2 cfg3; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$c -o $c --output-format csv -- python3 $R/bench.py --workload $c --steps 40 --warmup 4 --no-cpu-baseline > $O/prof_$c.log 2>&1
  cp $(find $O/prof_$c -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$c.csv
done
echo prof done
B="python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --phase-reps 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m --output-format csv -- $B > $O/pmc_mfma.log 2>&1
python3 $R/scripts/pmc_summary.py cfg4 $O/pmc_fetch $O/pmc_write $O/pmc_mfma > $O/pmc_cfg4.json
echo pmc done
python3 $R/scripts/run_configs.py cfg2 cfg3 cfg1 cfg4 > $O/configs_to_termination.jsonl 2> $O/configs.err
echo configs done
