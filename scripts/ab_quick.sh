#!/bin/bash
# Dev tool: bench lines + phase replay of a few configs on one box.  usage: scripts/ab_quick.sh <tag> [configs...]
R=$PWD
O=$R/gpurun_out/${1:-abq}
shift
mkdir -p $O
for c in ${*:-cfg5 cfg4 cfg2}; do
  for rep in 1 2; do
    timeout -k 10 300 python3 $R/bench.py --workload $c --steps 40 --warmup 5 --no-cpu-baseline --phase-reps 10 > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c rep=$rep value %.1f ms_per_step %.4f phases %s' % (r['value'], r['ms_per_step'], {k: round(v, 4) for k, v in r['phase_replay_ms'].items()}))" | tee -a $O/ab.log
  done
done
