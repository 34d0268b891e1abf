set -e
R=$PWD
O=$R/gpurun_out/r01m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 100 --warmup 10 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
for c in cfg1 cfg2 cfg3 cfg5; do python3 $R/bench.py --workload $c --steps 100 --warmup 10 > $O/bench_$c.json 2> $O/bench_$c.err; done
echo benches done
rocprofv3 --kernel-trace --stats -d $O/prof -o cfg4 --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/prof.log 2>&1
echo prof done
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --phase-reps 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --phase-reps 2 > $O/pmc_write.log 2>&1
echo pmc done
python3 $R/scripts/run_configs.py cfg2 cfg3 cfg1 cfg4 > $O/configs.log 2>&1
echo configs done
