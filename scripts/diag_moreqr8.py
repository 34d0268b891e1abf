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
k = 6
x = ro["snap"][k]; lam = ro["trace"][k, 5]
cam = x[:15 * N].copy(); pts = x[15 * N:].copy()
sq = ba.Solver(p, ba.QRCHOL, ba.F64); sq.set_state(cam.reshape(N, 15), pts); sq.linearize(False)
sm = ba.Solver(p, ba.MOREQR, ba.F64); sm.set_state(cam.reshape(N, 15), pts); sm.linearize(False)
for f in (1 - 1e-2, 1 - 1e-4, 1 - 1e-6, 1 - 1e-9, 1.0, 1 + 1e-12, 1 + 1e-9, 1 + 1e-6, 1 + 1e-4, 1.5, 0.3, 1.0):
    lm = lam * f
    sq.try_step(lm); a = sq.get(ba.GET_DX)[3 * M:]
    sm.try_step(lm); b = sm.get(ba.GET_DX)[3 * M:]
    print("lam x (1 %+.0e): MOREQR vs QRCHOL (both GPU) dx_c rel %.2e" % (f - 1, np.linalg.norm(a - b) / np.linalg.norm(a)))
# and a fresh linearisation in between
sm.linearize(False); sm.try_step(lam); b = sm.get(ba.GET_DX)[3 * M:]; sq.try_step(lam); a = sq.get(ba.GET_DX)[3 * M:]
print("after a second linearize, lam: %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(a)))
