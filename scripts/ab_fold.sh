#!/bin/bash
# Dev tool: same-box A/B of the folded launches (BA_NO_FOLD=1 = every launch of its own), two passes per config.
R=$PWD
O=$R/gpurun_out/${1:-ab}
mkdir -p $O
for rep in 1 2; do
  for c in cfg2 cfg1 cfg4 cfg5; do
    for nf in 0 1; do
      if [ $nf = 1 ]; then export BA_NO_FOLD=1; else unset BA_NO_FOLD; fi
      timeout -k 10 300 python3 $R/bench.py --workload $c --steps 100 --warmup 10 --no-cpu-baseline > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
      python3 -c "
import json,sys
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c nofold=$nf rep=$rep value %.1f ms_per_step %.4f' % (r['value'], r['ms_per_step']))" | tee -a $O/ab.log
    done
  done
done
