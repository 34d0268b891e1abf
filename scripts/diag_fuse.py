"""Dev diagnostic: which output of the fused linearisation differs from the separate launches (bits)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bundleadjustment_benchmarks_amd as ba
p = ba.Problem.load_bal(os.path.join(ROOT, "data", "problem-21-11315-pre.txt"))
out = {}
for nofuse in (0, 1):
    if nofuse: os.environ["BA_NO_FUSE"] = "1"
    else: os.environ.pop("BA_NO_FUSE", None)
    s = ba.Solver(p, ba.CHOLESKY, ba.F64)
    r = s.minimize(max_trials=2)
    out[nofuse] = dict(trace=r["trace"][:, :5], grad=s.get(ba.GET_GRAD), jc=s.get(ba.GET_JC), jp=s.get(ba.GET_JP), res=s.get(ba.GET_RESIDUALS), dx=s.get(ba.GET_DX))
for k in out[0]:
    a, b = out[0][k], out[1][k]
    print(k, "equal" if np.array_equal(a, b) else "DIFFERENT: %d of %d entries, max rel %.3e" % ((a != b).sum(), a.size, np.abs(a - b).max() / np.abs(b).max()))
print(out[0]["trace"]); print(out[1]["trace"])
