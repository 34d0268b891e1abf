import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
import oracle_lib as O
path = os.path.join(ROOT, "data", "problem-21-11315-pre.txt")
p = ba.Problem.load_bal(path); po = O.load_bal(path)
ro = O.minimize(O.MOREQR, po, max_trials=24, snapshots=True)
N = p.N; M = p.M
bad = 0
for kind in (ba.MOREQR, ba.QRKIT):
    sq = ba.Solver(p, ba.QRCHOL, ba.F64); sm = ba.Solver(p, kind, ba.F64)
    out = []
    for k in range(0, 24):
        x = ro["snap"][k]; lam = ro["trace"][k, 5]
        cam = x[:15 * N].copy(); pts = x[15 * N:].copy()
        sq.set_state(cam.reshape(N, 15), pts); sq.linearize(False); sq.try_step(lam); a = sq.get(ba.GET_DX)[3 * M:]
        sm.set_state(cam.reshape(N, 15), pts); sm.linearize(False); sm.try_step(lam); b = sm.get(ba.GET_DX)[3 * M:]
        sm.try_step(lam * (1 + 1e-12)); c = sm.get(ba.GET_DX)[3 * M:]
        out.append((k, lam, np.linalg.norm(a - b) / np.linalg.norm(a), np.linalg.norm(a - c) / np.linalg.norm(a)))
    print(ba.KIND_NAMES[kind], "BA_DBG_TAIL=%s" % os.environ.get("BA_DBG_TAIL"), " trial: rel(dx_c vs QRCHOL) exact lambda | lambda(1+1e-12)")
    for o in out: print("   %2d lam %.3e  %.2e | %.2e %s" % (o[0], o[1], o[2], o[3], "<-- " if o[2] > 30 * o[3] and o[2] > 1e-7 else ""))
