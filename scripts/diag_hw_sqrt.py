"""Diagnostic (VERDICT r3 item 8 i): config 3 (problem-39, QRKIT, fp32) with sqrtf and with the bare v_sqrt_f32 for the reflectors' beta."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bundleadjustment_benchmarks_amd as ba
p = ba.Problem.load_bal(os.path.join(ROOT, "data", "problem-39-18060-pre.txt"))
res = {}
for hw in (0, 1):
    os.environ["BA_QR_HW_SQRT"] = str(hw)
    s = ba.Solver(p, ba.QRKIT, ba.F32)
    e, dmax = s.linearize()
    lam = 1e-12 * dmax
    et, rs, dn = s.try_step(lam)
    rd = np.empty(p.D); ba._chk(ba.lib().ba_solver_get(s._h, 11, rd.ctypes.data_as(C.c_void_p), p.D), "get 11")
    dx = s.get(ba.GET_DX)
    r = s.minimize(max_trials=12)
    res[hw] = dict(e=e, et=et, dn=dn, rd=rd, dx=dx, trace=r["trace"])
    print("hw_sqrt=%d: energy %.6f  test energy %.6f  |dx| %.5g  accepted rows %d of %d" % (hw, e, et, dn, int(r["trace"][:, 1].sum()), len(r["trace"])))
    print(r["trace"][:, :5])
a, b = res[0], res[1]
rel = np.abs(a["rd"] - b["rd"]) / np.abs(a["rd"])
print("R diagonal: max rel diff %.3e at column %d; columns with rel diff > 1e-5: %s" % (rel.max(), int(rel.argmax()), np.where(rel > 1e-5)[0][:20]))
print("first 12 |R_jj| sqrtf  :", np.abs(a["rd"][:12]))
print("first 12 |R_jj| v_sqrt :", np.abs(b["rd"][:12]))
bad = np.where(~np.isfinite(b["rd"]))[0]
print("non-finite R_jj with v_sqrt at columns:", bad[:20], "count", len(bad))
j0 = int(bad[0]) if len(bad) else 0
print("R_jj sqrtf  around it:", a["rd"][max(0, j0 - 6): j0 + 4])
print("R_jj v_sqrt around it:", b["rd"][max(0, j0 - 6): j0 + 4])
print("smallest |R_jj| (sqrtf): %.3e at column %d; largest %.3e" % (np.abs(a["rd"]).min(), int(np.abs(a["rd"]).argmin()), np.abs(a["rd"]).max()))
print("dx camera part rel diff %.3e" % (np.linalg.norm(a["dx"][3 * p.M:] - b["dx"][3 * p.M:]) / np.linalg.norm(a["dx"][3 * p.M:])))
