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
for k in (5, 6):
    x = ro["snap"][k]; lam = ro["trace"][k, 5]
    cam = x[:15 * N].copy(); pts = x[15 * N:].copy()
    f, e = O.residuals(po, cam, pts); Jc, Jp = O.jacobian(po, cam, pts)
    st = O.step(O.QRCHOL, po, Jc, Jp, f, lam, want_S=False)
    for kind in (ba.QRKIT, ba.MOREQR, ba.QRCHOL):
        s = ba.Solver(p, kind, ba.F64)
        s.set_state(cam.reshape(N, 15), pts); s.linearize(False)
        for lm in (lam, lam * 1.0001, lam * 3):
            et, rs, dn = s.try_step(lm)
            sto = st if lm == lam else O.step(O.QRCHOL, po, Jc, Jp, f, lm, want_S=False)
            dx = s.get(ba.GET_DX)
            msg = "trial %d kind %s lam %.6e: dxc rel vs oracle QRCHOL %.2e" % (k, ba.KIND_NAMES[kind], lm, np.linalg.norm(dx[3 * M:] - sto["dx"][3 * M:]) / np.linalg.norm(sto["dx"][3 * M:]))
            if kind != ba.QRCHOL:
                rd = np.empty(p.D); ba._chk(ba.lib().ba_solver_get(s._h, 11, rd.ctypes.data_as(C.c_void_p), p.D), "get 11")
                msg += "  min |R_jj| %.3e at %d, max %.3e" % (np.abs(rd).min(), int(np.abs(rd).argmin()), np.abs(rd).max())
            print(msg)
