#!/bin/bash
# Dev tool: one gpurun call = dense bench (backward sweep packed onto one XCD or not), two bench lines, the GPU test suite.
# usage: scripts/gpu_round.sh <tag> [pytest -k expression]
R=$PWD
O=$R/gpurun_out/${1:-round}
mkdir -p $O
set -o pipefail
for D in 189 2313 9216; do
  for P in 0 1; do BA_SWEEP_PACK=$P timeout -k 10 120 $R/scripts/bench_dense.bin $D 2>&1 | grep -E "^fused|^D=" | sed "s/^/pack=$P /" >> $O/dense.log || exit 1; done
done
cat $O/dense.log
timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err || { tail -5 $O/bench_cfg4.err; exit 1; }
timeout -k 10 300 python3 $R/bench.py --workload cfg2 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err || { tail -5 $O/bench_cfg2.err; exit 1; }
BA_SWEEP_PACK=1 timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_pack.json 2> $O/bench_cfg4_pack.err || { tail -5 $O/bench_cfg4_pack.err; exit 1; }
python3 - $O <<'PY'
import json, sys
for n in ("bench_cfg4", "bench_cfg2", "bench_cfg4_pack"):
    r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
    print(n, r["value"], r["ms_per_step"], r.get("phase_replay_ms"))
PY
cd $R && timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ${2:+-k "$2"} > $O/pytest.log 2>&1
rc=$?
tail -15 $O/pytest.log
exit $rc
