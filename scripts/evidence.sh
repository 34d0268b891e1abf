#!/bin/bash
# Round evidence, one gpurun call: bench lines of every config, rocprofv3 kernel stats, PMC passes (HBM bytes: FETCH_SIZE and
# WRITE_SIZE in passes of their own; matrix-core busy cycles) for configs 4, 5 and 3, the FETCH_SIZE control of scripts/fetch_control.hip,
# the configs run to the reference's own stop next to the oracle (config 4 included) with the quad free run / ensemble of problem-21.
# usage: scripts/evidence.sh <tag> [part ...]  (parts: bench prof pmc control configs; default all)  -> gpurun_out/<tag>/ ;
# copy what is to be judged into profiles/ (scripts/evidence_copy.sh <tag>).
R=$PWD
O=$R/gpurun_out/${1:-r04ev}
shift
PARTS=${*:-bench prof pmc control configs}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python3 $R/bench.py --steps 100 --warmup 10 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
  for c in cfg1 cfg2 cfg3 cfg5; do python3 $R/bench.py --workload $c --steps 100 --warmup 10 > $O/bench_$c.json 2> $O/bench_$c.err; done
  echo benches done
fi
if has prof; then
  for c in cfg4 cfg5 cfg2 cfg3; do
    rocprofv3 --kernel-trace --stats -d $O/prof_$c -o $c --output-format csv -- python3 $R/bench.py --workload $c --steps 40 --warmup 4 --no-cpu-baseline > $O/prof_$c.log 2>&1
    cp $(find $O/prof_$c -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$c.csv
    rm -rf $O/prof_$c
  done
  echo prof done
fi
if has pmc; then
  for c in cfg4 cfg5 cfg3; do
    B="python3 $R/bench.py --workload $c --steps 20 --warmup 2 --regions 1 --no-cpu-baseline --phase-reps 2"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$c -o f --output-format csv -- $B > $O/pmc_fetch_$c.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$c -o w --output-format csv -- $B > $O/pmc_write_$c.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/pmc_mfma_$c -o m --output-format csv -- $B > $O/pmc_mfma_$c.log 2>&1
    python3 $R/scripts/pmc_summary.py $c $O/pmc_fetch_$c $O/pmc_write_$c $O/pmc_mfma_$c > $O/pmc_$c.json
    rm -rf $O/pmc_fetch_$c $O/pmc_write_$c $O/pmc_mfma_$c # (per-dispatch counter tables: tens of MB; gpurun brings back 64 MiB)
    echo pmc $c done
  done
fi
if has control; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fc -o fc --output-format csv -- $R/scripts/fetch_control.bin > $O/fetch_control.log 2>&1
  echo control done
fi
if has configs; then
  python3 $R/scripts/run_configs.py ${CFGS:-cfg2 cfg3 cfg1 cfg4} > $O/configs_to_termination.jsonl 2> $O/configs.err
  echo configs done
fi
