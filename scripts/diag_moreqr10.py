import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
import oracle_lib as O
path = os.path.join(ROOT, "data", "problem-21-11315-pre.txt")
p = ba.Problem.load_bal(path); po = O.load_bal(path)
ro = O.minimize(O.MOREQR, po, max_trials=8, snapshots=True)
N = p.N; M = p.M
x = ro["snap"][6]; lam = ro["trace"][6, 5]
cam = x[:15 * N].copy(); pts = x[15 * N:].copy()
sq = ba.Solver(p, ba.QRCHOL, ba.F64); sq.set_state(cam.reshape(N, 15), pts); sq.linearize(False); sq.try_step(lam); a = sq.get(ba.GET_DX)[3 * M:]
sm = ba.Solver(p, ba.MOREQR, ba.F64); sm.set_state(cam.reshape(N, 15), pts); sm.linearize(False); sm.try_step(lam); b = sm.get(ba.GET_DX)[3 * M:]
print("BA_DBG_TAIL=%s: MOREQR vs QRCHOL dx_c rel %.2e" % (os.environ.get("BA_DBG_TAIL"), np.linalg.norm(a - b) / np.linalg.norm(a)))
