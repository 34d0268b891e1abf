"""Diagnostic: self-check |A^T (b - A y)| / |A^T b| of every dense least-squares solve over many lambdas and states (BA_DBG_QRCHECK=1)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
import oracle_lib as O
path = os.path.join(ROOT, "data", "problem-21-11315-pre.txt")
p = ba.Problem.load_bal(path); po = O.load_bal(path)
ro = O.minimize(O.QRCHOL, po, max_trials=30, snapshots=True)
N = p.N; M = p.M; D = p.D
def check(s):
    r = np.empty(2 * D); ba._chk(ba.lib().ba_solver_get(s._h, 14, r.ctypes.data_as(C.c_void_p), 2 * D), "get 14")
    return np.linalg.norm(r[:D]) / np.linalg.norm(r[D:])
rng = np.random.default_rng(0)
for kind in [getattr(ba, k) for k in os.environ.get('KINDS', 'QRKIT').split()]: # (KINDS='QRKIT MOREQR')
    s = ba.Solver(p, kind, ba.F64)
    vals = []
    for k in (0, 4, 8, 12, 16, 20, 24, 28):
        x = ro["snap"][k]
        s.set_state(x[:15 * N].reshape(N, 15), x[15 * N:]); s.linearize(False)
        for lam0 in (1e-1, 1e-3, 1e-5, 1e-7, 1e-9):
            for rep in range(8):
                lam = lam0 * (1 + 1e-3 * rng.standard_normal())
                s.try_step(lam)
                vals.append((k, lam, check(s)))
    v = np.array([w[2] for w in vals])
    print(ba.KIND_NAMES[kind], "solves", len(v), "self-check |A'(b - Ay)|/|A'b|: median %.2e  90%% %.2e  max %.2e" % (np.median(v), np.quantile(v, 0.9), v.max()))
    bad = [w for w in vals if w[2] > 50 * np.median(v)]
    print("   outliers (> 50 x median):", len(bad), [("%d" % w[0], "%.6e" % w[1], "%.1e" % w[2]) for w in bad[:12]])
