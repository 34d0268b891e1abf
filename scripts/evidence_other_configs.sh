set -e
R=$PWD
O=$R/gpurun_out/r01n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cfg5 cfg2 cfg3; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$c -o $c --output-format csv -- python3 $R/bench.py --workload $c --steps 40 --warmup 4 --no-cpu-baseline > $O/prof_$c.log 2>&1
  echo $c done
done
