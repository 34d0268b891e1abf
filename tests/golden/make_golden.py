"""Regenerates tests/golden/*.json from the CPU oracle and the two BAL files the reference ships.

The reference has no tests, golden vectors or recorded output and cannot be built offline, so these
fixtures pin the ORACLE (regression) -- "parity unpinned" against the reference itself.  Independent pins kept beside
them: SURVEY.md 6.2 (the surveyor's separate NumPy/SciPy prototype), tests/test_oracle_math.py (finite differences,
scipy sparse solve).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

SAMPLE_OBS = [0, 1, 2, 17, 1000, 5000, 20000, 36454]


def make(name, fname, ntr):
    p = O.load_bal(os.path.join(ROOT, "data", fname))
    cam = O.init_cams(p)
    f, e = O.residuals(p, cam, p.pts)
    Jc, Jp = O.jacobian(p, cam, p.pts)
    st = O.stats(p, cam, p.pts)
    step = O.step(O.CHOLESKY, p, Jc, Jp, f, 1.0, want_S=False)
    lam0 = 1e-12 * step["diagmax"]
    out = {
        "file": fname, "N": p.N, "M": p.M, "K": p.K,
        "energy0": e, "stats0": {k: float(v) for k, v in st.items()},
        "grad_norm0": float(np.linalg.norm(step["g"])), "diag_max0": step["diagmax"], "lambda0": lam0,
        "sample_obs": [i for i in SAMPLE_OBS if i < p.K],
    }
    idx = out["sample_obs"]
    out["sample_residuals"] = f.reshape(-1, 2)[idx].tolist()
    out["sample_Jc"] = Jc[idx].tolist()
    out["sample_Jp"] = Jp[idx].tolist()
    for kind, kname in ((O.CHOLESKY, "CHOLESKY"), (O.QRCHOL, "QRCHOL")):
        s1 = O.step(kind, p, Jc, Jp, f, lam0, want_S=True)
        out["first_step_" + kname] = {"dx_norm": float(np.linalg.norm(s1["dx"])), "S_fro": float(np.linalg.norm(s1["S"])),
                                      "S_trace": float(np.trace(s1["S"])), "rhs_norm": float(np.linalg.norm(s1["rhs"]))}
        r = O.minimize(kind, p, max_trials=ntr)
        out["trace_" + kname] = {"columns": list(O.TRACE_COLS), "rows": r["trace"].tolist(), "status": r["status"]}
    with open(os.path.join(HERE, name + ".json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(name, "energy0", e, "lambda0", lam0)


if __name__ == "__main__":
    make("oracle_problem21", "problem-21-11315-pre.txt", 12)
    make("oracle_problem39", "problem-39-18060-pre.txt", 12)
