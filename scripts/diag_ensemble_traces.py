"""Dev diagnostic: LM tables (iter, accepted, f, rho, lambda, elapsed) of the GPU's free runs on ensemble members 0..n-1 -> npz."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
import ensemble_lib as E
import oracle_lib as O
case, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kind = getattr(ba, E.CASES[case][1])
p0 = E.base_problem(ba, O, case)
d = {}
for k in range(n):
    po = O.ensemble_member(p0, k)
    pg = ba.Problem.from_arrays(po.N, po.M, po.K, po.cam_idx, po.pt_idx, po.meas, po.cams9, po.pts)
    s = ba.Solver(pg, kind, ba.F64)
    r = s.minimize()
    d["trace%d" % k] = r["trace"]
np.savez_compressed(out, **d)
