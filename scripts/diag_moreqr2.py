"""Dev diagnostic: after 3 production trials, the 4th through the seam; and the production run with BA_NO_FUSE / BA_QR_ONE_STREAM."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bundleadjustment_benchmarks_amd as ba
p = ba.Problem.load_bal(os.path.join(ROOT, "data", "problem-21-11315-pre.txt"))
s = ba.Solver(p, ba.MOREQR, ba.F64)
r = s.minimize(max_trials=3)
lam = r["lam"]
print("after 3 trials: energy %.9f lambda %.6e" % (r["energy"], lam))
e, _ = s.linearize(False)
et, rs, dn = s.try_step(lam)
print("4th trial through the seam from the production state: energy %.9f e_test %.9f (oracle 1620.452841405)" % (e, et))
for env in ({"BA_NO_FUSE": "1"}, {"BA_QR_ONE_STREAM": "1"}, {"BA_NO_FOLD": "1"}):
    for k in ("BA_NO_FUSE", "BA_QR_ONE_STREAM", "BA_NO_FOLD"): os.environ.pop(k, None)
    os.environ.update(env)
    s2 = ba.Solver(p, ba.MOREQR, ba.F64)
    r2 = s2.minimize(max_trials=6)
    print(env, "f:", r2["trace"][:, 2])
