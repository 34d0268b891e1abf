"""GPU side of the 1e-13 ensembles: the perturbed inputs of tests/golden/referee_ensemble_*.json through ba_minimize.

    python scripts/gpu_ensemble.py [n_members] > gpurun_out/r04_final_cost_ensembles.json      (on the GPU box)

Prints one JSON document: per case the GPU members (status, trials, final energy by the oracle's residual function), the oracle's
members from the fixture, and the distribution comparison (medians, ranges, two-sided Mann-Whitney p on final energy and trials).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba  # noqa: E402
import ensemble_lib as E  # noqa: E402
import oracle_lib as O  # noqa: E402  (checker: the final energies of both sides are evaluated by the oracle's residual function)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
doc = {}
for case in E.CASES:
    ens = E.fixture(case)
    gpu = E.gpu_members(ba, O, case, n)
    doc[case] = dict(compare=E.compare(ens, gpu),
                     compare_with_oracle_accurate_S_sums=E.compare(E.fixture(case + "_widesums"), gpu),
                     compare_with_oracle_in_x87_long_double=E.compare(E.fixture(case + "_x87"), gpu),
                     gpu_members=[dict(member=g[0], status=g[1], trials=g[2], final_energy=g[3], iterations=g[4]) for g in gpu],
                     oracle_members=ens["members"][:n])
    for k in ("compare", "compare_with_oracle_accurate_S_sums", "compare_with_oracle_in_x87_long_double"):
        print(case, k, json.dumps(doc[case][k]), file=sys.stderr, flush=True)
print(json.dumps(doc, indent=1))
