"""The 1e-13 ensembles on the GPU side -- shared by tests/test_gpu_configs.py and scripts/gpu_ensemble.py.

tests/golden/referee_ensemble_<case>.json holds, per member k, the fp64 ORACLE's free run to the reference's own stop from the
input perturbed by 1e-13 (oracle_lib.ensemble_member).  gpu_members() runs the SAME perturbed inputs through ba_minimize; the
final energy of either side is the oracle's residual function applied to the state the loop leaves behind (one yardstick).
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CASES = {
    # fixture name: (source, solver symbol)
    "problem21_qrchol": (("bal", "problem-21-11315-pre.txt"), "QRCHOL"),
    "problem21_cholesky": (("bal", "problem-21-11315-pre.txt"), "CHOLESKY"),
    "cfg1_cholesky": (("synthetic", (16, 22106, 83718, 1001)), "CHOLESKY"),
}


def fixture(case):
    return json.load(open(os.path.join(HERE, "golden", "referee_ensemble_%s.json" % case)))


def base_problem(ba, O, case):
    source = CASES[case][0]
    if source[0] == "bal":
        return O.load_bal(os.path.join(ROOT, "data", source[1]))
    a = ba.Problem.synthetic(*source[1])
    arr = a.arrays()
    return O.Problem(a.N, a.M, a.K, arr["cam_idx"], arr["pt_idx"], arr["meas"], arr["cams9"], arr["pts"])


def gpu_members(ba, O, case, n):
    """[(member, status, trials, final_energy, iterations)] of the GPU's free runs on members 0 .. n-1."""
    kind = getattr(ba, CASES[case][1])
    p0 = base_problem(ba, O, case)
    out = []
    for k in range(n):
        po = O.ensemble_member(p0, k)
        pg = ba.Problem.from_arrays(po.N, po.M, po.K, po.cam_idx, po.pt_idx, po.meas, po.cams9, po.pts)
        s = ba.Solver(pg, kind, ba.F64)
        r = s.minimize(trace=False)
        _, e = O.residuals(po, s.get(ba.GET_CAMS), s.get(ba.GET_POINTS))
        out.append((k, int(r["status"]), int(r["trials"]), float(e), int(r["iterations"])))
        del s, pg
    return out


def mann_whitney_p(a, b):
    """Two-sided Mann-Whitney U test (normal approximation with tie correction; scipy when importable)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    try:
        from scipy.stats import mannwhitneyu
        return float(mannwhitneyu(a, b, alternative="two-sided").pvalue)
    except ImportError:
        pass
    from math import erfc, sqrt
    n1, n2 = len(a), len(b)
    allv = np.concatenate([a, b])
    order = allv.argsort(kind="mergesort")
    ranks = np.empty(n1 + n2)
    sv = allv[order]
    i = 0
    tie = 0.0
    while i < len(sv):
        j = i
        while j + 1 < len(sv) and sv[j + 1] == sv[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1
        t = j - i + 1
        tie += t ** 3 - t
        i = j + 1
    u = ranks[:n1].sum() - n1 * (n1 + 1) / 2
    n = n1 + n2
    sd = sqrt(n1 * n2 / 12.0 * ((n + 1) - tie / (n * (n - 1))))
    z = (abs(u - n1 * n2 / 2.0) - 0.5) / sd
    return float(erfc(z / sqrt(2)))


def compare(ens, gpu):
    """Distribution-level comparison of the oracle ensemble (fixture dict) with the GPU members (gpu_members())."""
    n = len(gpu)
    mem = [m for m in ens["members"] if m["member"] < n]
    eo = np.array([m["final_energy"] for m in mem])
    to = np.array([m["trials"] for m in mem])
    eg = np.array([g[3] for g in gpu])
    tg = np.array([g[2] for g in gpu])
    return dict(n=n, oracle_energy_min=float(eo.min()), oracle_energy_max=float(eo.max()), oracle_energy_median=float(np.median(eo)),
                gpu_energy_min=float(eg.min()), gpu_energy_max=float(eg.max()), gpu_energy_median=float(np.median(eg)),
                oracle_trials_min=int(to.min()), oracle_trials_max=int(to.max()), oracle_trials_median=float(np.median(to)),
                gpu_trials_min=int(tg.min()), gpu_trials_max=int(tg.max()), gpu_trials_median=float(np.median(tg)),
                p_energy=mann_whitney_p(eg, eo), p_trials=mann_whitney_p(tg, to),
                gpu_all_success=bool(all(g[1] == 0 for g in gpu)), oracle_all_success=bool(all(m["status"] == 0 for m in mem)))
