"""Dev diagnostic: tests/test_gpu_referee.measure() rows (per trial, oracle's states injected) -> npz, for a signed look at e_test - e_quad."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
import oracle_lib as O
import test_gpu_referee as R
out = sys.argv[1]
d = {}
for name in sys.argv[2:]:
    m = R.measure(ba, O, name)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "referee_%s.json" % name)))
    d[name] = m
    d[name + "_efp64"] = np.array([w["e_test_fp64"] for w in fx["trials"]])
np.savez_compressed(out, **d)
