"""Diagnostic: the GPU's own fp32 QRKIT trajectory on problem-39 through the step-level seam (the loop of tests/test_gpu_referee.py);
prints, per trial, lambda, the test energy and whether step / state hold non-finite values."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bundleadjustment_benchmarks_amd as ba
p = ba.Problem.load_bal(os.path.join(ROOT, "data", "problem-39-18060-pre.txt"))
z = np.load(os.path.join(ROOT, "tests", "golden", "referee_problem39_qrkit_f32_states.npz"))
N = p.N
s = ba.Solver(p, ba.QRKIT, ba.F32)
x0 = z["states"][0].astype(np.float64)
s.set_state(x0[: 15 * N].reshape(N, 15), x0[15 * N:])
e, dmax = s.linearize()
lam, inc = float(np.float32(1e-12 * dmax)), 2.0
for k in range(16):
    et, rs, dn = s.try_step(lam)
    dx = s.get(ba.GET_DX)
    bad = np.nonzero(~np.isfinite(dx))[0]
    xt = np.concatenate([s.get(ba.GET_CAMS_TEST).ravel(), s.get(ba.GET_POINTS_TEST).ravel()])
    print("k=%2d lam %.4e e %.6e et %.6e rs %.3e |dx| %.3e  nonfinite dx %d (first %s; 3M=%d)  nonfinite xTest %d  max|dx| %.3e" % (
        k, lam, e, et, rs, dn, len(bad), bad[:5], 3 * p.M, int((~np.isfinite(xt)).sum()), np.nanmax(np.abs(dx))))
    if et < e:
        rho = (e - et) / rs
        lam = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), 1e-10)
        inc = 2.0
        s.accept()
        e, _ = s.linearize(False)
    else:
        lam *= inc
        inc = inc ** 1.5
    lam = float(np.float32(lam))
