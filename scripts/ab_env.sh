#!/bin/bash
# Dev tool: bench line + phase replay of one config under a list of environment settings, on one box.
# usage: scripts/ab_env.sh <tag> <config> "VAR=a VAR2=b" "VAR=c" ...
R=$PWD
O=$R/gpurun_out/${1:-abe}
C=$2
shift 2
mkdir -p $O
for e in "" "$@"; do
    env $e timeout -k 10 300 python3 $R/bench.py --workload $C --steps 30 --warmup 5 --no-cpu-baseline --phase-reps 10 > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$C [$e] value %.1f ms_per_step %.4f schur %.4f' % (r['value'], r['ms_per_step'], r['phase_replay_ms']['schur_assembly']))" | tee -a $O/ab.log
done
