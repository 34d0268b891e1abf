"""Dev diagnostic: LM driven from the host through the step-level seam on the GPU; per trial e, e_test, |dx|, lambda, state norms."""
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
    rows = []
    lam = None; inc = 2.0; hist = [0.0, 0.0]; it = 0; stop = False
    while not stop and len(rows) < 600:
        it += 1
        e, dmax = s.linearize(True)
        if it == 1: lam = 1e-12 * dmax
        cams = s.get(ba.GET_CAMS).reshape(15, -1) if False else s.get(ba.GET_CAMS)
        pts = s.get(ba.GET_POINTS)
        while True:
            et, rs, dn = s.try_step(lam)
            if et < e:
                rho = (e - et) / rs
                lam_used = lam
                lam = max(lam * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3), 1e-10)
                rows.append((it, 1, e, rho, lam, lam_used, et, dn, np.linalg.norm(pts), np.linalg.norm(cams)))
                inc = 2.0; e_new = et; hist[it % 2] = e_new
                break
            else:
                rows.append((it, 0, e, 0, lam, lam, et, dn, np.linalg.norm(pts), np.linalg.norm(cams)))
                if lam > 1e10: stop = True; break
                lam *= inc; inc = inc ** 1.5
        if stop: break
        if it > 2 and abs(e_new - max(hist)) < 1e-8 * e_new: break
        s.accept()
    d["rows%d" % k] = np.array(rows)
np.savez_compressed(out, **d)
