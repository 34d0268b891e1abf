#!/bin/bash
# One gpurun call: GPU test suite (all failures shown), then the default bench line and a phase breakdown.
# usage: scripts/gpu_check.sh <tag> [pytest -k expression]
set -o pipefail
R=$PWD
TAG=${1:-check}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout -k 10 1500 python3 -m pytest $R/tests -m gpu -q -rA --durations=12 -p no:cacheprovider "${K[@]}" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout -k 10 300 python3 $R/bench.py --steps 50 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
echo "bench rc=$?"
python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1])
    print("cfg4: %.1f it/s  ms/step %.4f  schur_solve %.4f  phases %s" % (d["value"], d["ms_per_step"], d["schur_solve_ms"], d.get("phase_replay_ms")))
except Exception as e:
    print("bench parse failed:", e)
PY
