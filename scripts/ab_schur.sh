#!/bin/bash
# Dev tool: same-box A/B of the Schur assembly's chunk dealing (BA_SCHUR_WINDOW = chunks per wavefront and window, 0 = one window).
R=$PWD
O=$R/gpurun_out/${1:-abs}
mkdir -p $O
for c in cfg5 cfg4 cfg2; do
  for k in 0 1 2 4 8 0; do
    BA_SCHUR_WINDOW=$k timeout -k 10 300 python3 $R/bench.py --workload $c --steps 40 --warmup 5 --no-cpu-baseline --phase-reps 10 > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c window=$k value %.1f ms_per_step %.4f schur %.4f ms' % (r['value'], r['ms_per_step'], r['phase_replay_ms']['schur_assembly']))" | tee -a $O/ab.log
  done
done
