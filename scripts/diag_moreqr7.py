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
k = 6
x = ro["snap"][k]; lam = ro["trace"][k, 5]
cam = x[:15 * N].copy(); pts = x[15 * N:].copy()
def rdiag(s):
    rd = np.empty(p.D); ba._chk(ba.lib().ba_solver_get(s._h, 11, rd.ctypes.data_as(C.c_void_p), p.D), "get 11"); return rd
res = {}
for kind in (ba.QRKIT, ba.MOREQR):
    s = ba.Solver(p, kind, ba.F64)
    s.set_state(cam.reshape(N, 15), pts); s.linearize(False)
    for lm in (lam, lam * 1.0001):
        s.try_step(lm)
        res[(kind, lm)] = (np.abs(rdiag(s)), s.get(ba.GET_DX)[3 * M:].copy())
a = res[(ba.QRKIT, lam)]; b = res[(ba.MOREQR, lam)]; c = res[(ba.MOREQR, lam * 1.0001)]
rel = np.abs(a[0] - b[0]) / a[0]
print("MOREQR(lam) vs QRKIT(lam): |R_jj| rel diff max %.3e at col %d; cols > 1e-8: %s" % (rel.max(), int(rel.argmax()), np.where(rel > 1e-8)[0][:30]))
rel2 = np.abs(c[0] - b[0]) / c[0]
print("MOREQR(lam) vs MOREQR(lam*1.0001): |R_jj| rel diff max %.3e at col %d; cols > 1e-3: %s" % (rel2.max(), int(rel2.argmax()), np.where(rel2 > 1e-3)[0][:30]))
d = b[1] - a[1]
print("dx_c diff MOREQR - QRKIT by camera param index (mean over cameras):", np.abs(d.reshape(N, 9)).mean(axis=0))
print("dx_c diff largest entries:", np.argsort(-np.abs(d))[:12], np.sort(-np.abs(d))[:6])
