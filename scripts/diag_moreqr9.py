import ctypes as C, os, sys
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
def rdiag(s):
    rd = np.empty(p.D); ba._chk(ba.lib().ba_solver_get(s._h, 11, rd.ctypes.data_as(C.c_void_p), p.D), "get 11"); return rd
sm = ba.Solver(p, ba.MOREQR, ba.F64); sm.set_state(cam.reshape(N, 15), pts); sm.linearize(False)
sm.try_step(lam); r0 = rdiag(sm); d0 = sm.get(ba.GET_DX)
sm.try_step(lam * (1 + 1e-12)); r1 = rdiag(sm); d1 = sm.get(ba.GET_DX)
rel = np.abs(np.abs(r0) - np.abs(r1)) / np.abs(r1)
np.set_printoptions(linewidth=220)
print("columns where |R_jj| differs by > 1e-9 between lam and lam(1+1e-12):", np.where(rel > 1e-9)[0])
print("their rel diffs:", rel[rel > 1e-9])
print("sign flips of R_jj:", np.where(np.sign(r0) != np.sign(r1))[0])
dp = (d0[:3 * M] - d1[:3 * M]).reshape(M, 3)
print("point step diff: mean %s  std %s" % (dp.mean(axis=0), dp.std(axis=0)))
print("sqrt(lam) = %.17g ; R_jj values near it:" % np.sqrt(lam), r0[np.abs(np.abs(r0) - np.sqrt(lam)) < 1e-3 * np.sqrt(lam)])
