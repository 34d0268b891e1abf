"""Dev diagnostic: MOREQR (QR only) free run on the GPU, graphs on / off, against the oracle's first rows."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
import oracle_lib as O
p = ba.Problem.load_bal(os.path.join(ROOT, "data", "problem-21-11315-pre.txt"))
po = O.load_bal(os.path.join(ROOT, "data", "problem-21-11315-pre.txt"))
ro = O.minimize(O.MOREQR, po, max_trials=6, snapshots=True)
print("oracle f:", ro["trace"][:, 2])
for nog in (0, 1):
    if nog: os.environ["BA_NO_GRAPH"] = "1"
    else: os.environ.pop("BA_NO_GRAPH", None)
    s = ba.Solver(p, ba.MOREQR, ba.F64)
    r = s.minimize(max_trials=6)
    print("gpu no_graph=%d f:" % nog, r["trace"][:, 2])
# seam: the oracle's states injected
s = ba.Solver(p, ba.MOREQR, ba.F64)
N = p.N
for k in range(6):
    x = ro["snap"][k]; lam = ro["trace"][k, 5]
    s.set_state(x[:15 * N].reshape(N, 15), x[15 * N:])
    e, _ = s.linearize(False)
    et, rs, dn = s.try_step(lam)
    print("trial %d lam %.3e: gpu e_test %.9f oracle %.9f  rel %.2e" % (k, lam, et, ro["trace"][k, 6], abs(et - ro["trace"][k, 6]) / et))
np.set_printoptions(linewidth=200, precision=10)
s = ba.Solver(p, ba.MOREQR, ba.F64)
r = s.minimize(max_trials=6)
print("gpu rows (iter, acc, f, rho, lambda):\n", r["trace"][:, :5])
print("oracle rows (iter, acc, f, rho, lambda printed, lambda used, e_test):\n", ro["trace"][:, :7])
