"""Generates tests/golden/referee_*.json: per-trial quad-precision (__float128) values along the fp64 oracle's LM trajectory.

    python tests/golden/make_referee.py [case ...]        (about 10 minutes on 8 cores for all cases)

For every trial k of the oracle's free run (to the reference's own stop, at most MAX_TRIALS rows) the state x_k the trial
starts from and its lambda_k are handed, as doubles, to oracle/ba_referee.c, which evaluates the whole trial in 113-bit
arithmetic: energy, step by the solver symbol's elimination, retraction, test energy, rho denominator.  The fixture keeps
the oracle's own fp64 numbers beside the quad ones, so that (a) a test can check that its replay of the oracle is on the
fixture's trajectory (energy column, bit-sensitive), and (b) the fp64 oracle's distance from the truth is on record for
every lambda regime, including lambda < 1e-9 where cond(J'J + lambda I) > 1e19.

The inputs are the two BAL files the reference ships (copied to data/) and one seeded synthetic problem; nothing of the
reference's code is executed (it cannot be built here).
"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as O  # noqa: E402

MAX_TRIALS = 260
CASES = {
    # name: (source, solver kind[, trials])
    "problem21_qrchol": (("bal", "problem-21-11315-pre.txt"), O.QRCHOL),
    "problem21_cholesky": (("bal", "problem-21-11315-pre.txt"), O.CHOLESKY),
    # (QR-only right block on both sides since round 4: an oracle trial is two dense Householder QRs, ~2 s; 120 rows reach the lambda floor)
    # The quad trial eliminates as QRCHOL does: the step is the same least-squares solution whatever the route, and a __float128 dense QR
    # of the 107 044 x 189 J2bot would take minutes per trial.
    "problem21_moreqr": (("bal", "problem-21-11315-pre.txt"), O.MOREQR, 120, np.float64, O.QRCHOL),
    "problem39_qrchol": (("bal", "problem-39-18060-pre.txt"), O.QRCHOL),
    "synthetic60_cholesky": (("synthetic", (60, 3000, 14000, 2060)), O.CHOLESKY),
    # the headline workload (config 4 stand-in): a quad trial costs ~5 CPU-minutes (dense LDL^T of 2313^2 in __float128),
    # so only the first rows -- the ones the free-running prefix test of this config looks at
    "cfg4_cholesky": (("synthetic", (257, 65132, 225911, 1004)), O.CHOLESKY, 8),
    "cfg1_cholesky": (("synthetic", (16, 22106, 83718, 1001)), O.CHOLESKY, 24),
    # config 3 (QRKIT, Scalar = float): the FP32 oracle's own free run to its stop (28 trials, ExceededLambdaMax); the states are
    # floats, handed to the referee exactly (the first 24 trials: 17 states, 3.7 MB).  The quad trial uses the QRCHOL elimination: the step is the same least-squares
    # solution whatever the symbol, and in 113 bits the normal equations lose nothing that matters (a quad dense QR of the
    # 181 633 x 351 J2bot would take ~20 minutes per trial).  The states themselves are kept too (referee_<case>_states.npz,
    # float32, one per outer iteration): replaying the fp32 oracle's dense QR at test time would take 18 CPU-seconds per trial.
    "problem39_qrkit_f32": (("bal", "problem-39-18060-pre.txt"), O.QRKIT, 24, np.float32, O.QRCHOL),
}


def load(source):
    if source[0] == "bal":
        return O.load_bal(os.path.join(ROOT, "data", source[1]))
    import bundleadjustment_benchmarks_amd as ba  # host-side generator only (no GPU call)
    a = ba.Problem.synthetic(*source[1])
    arr = a.arrays()
    return O.Problem(a.N, a.M, a.K, arr["cam_idx"], arr["pt_idx"], arr["meas"], arr["cams9"], arr["pts"])


_P = {}


def _one(args):
    name, kind, k, state, lam = args
    if name not in _P:
        _P[name] = load(CASES[name][0])
    p = _P[name]
    rkind = CASES[name][4] if len(CASES[name]) > 4 else kind  # (the symbol the quad trial eliminates with)
    q = O.referee_trial(rkind, p, state[: 15 * p.N], state[15 * p.N:], lam)
    return k, q


def make(name, pool):
    source, kind = CASES[name][:2]
    max_trials = CASES[name][2] if len(CASES[name]) > 2 else MAX_TRIALS
    dtype = CASES[name][3] if len(CASES[name]) > 3 else np.float64
    p = load(source)
    r = O.minimize(kind, p, dtype=dtype, max_trials=max_trials, snapshots=True)
    tr, sn = r["trace"], r["snap"]
    if dtype == np.float32:  # the states (exact floats), one per outer iteration, + the map trial -> state
        it = tr[:, 0].astype(int)
        first = [int(np.argmax(it == v)) for v in np.unique(it)]
        np.savez_compressed(os.path.join(HERE, "referee_%s_states.npz" % name), states=sn[first].astype(np.float32),
                            state_of_trial=np.searchsorted(np.unique(it), it).astype(np.int32))
    jobs = [(name, kind, k, sn[k], float(tr[k][5])) for k in range(len(tr))]
    res = dict(pool.imap_unordered(_one, jobs, chunksize=1))
    rows = []
    for k in range(len(tr)):
        q = res[k]
        rows.append(dict(trial=k, iter=int(tr[k][0]), accepted=int(tr[k][1]), lam=float(tr[k][5]),
                         energy_fp64=float(tr[k][2]), e_test_fp64=float(tr[k][6]), dx_norm_fp64=float(tr[k][7]),
                         rho_fp64=float(tr[k][3]),
                         energy_quad=q["energy"], e_test_quad=q["e_test"], rho_scale_quad=q["rho_scale"],
                         dx_norm_quad=q["dx_norm"], backward_error_quad=q["backward_error"]))
    out = dict(case=name, source=list(source[:1]) + [source[1] if isinstance(source[1], str) else list(source[1])], kind=int(kind),
               N=p.N, M=p.M, K=p.K, status=int(r["status"]), max_trials=max_trials, scalar="f32" if dtype == np.float32 else "f64",
               note="fp64 columns: oracle/ba_oracle.c free run; quad columns: oracle/ba_referee.c from the same (x, lambda) per trial",
               trials=rows)
    with open(os.path.join(HERE, "referee_%s.json" % name), "w") as f:
        json.dump(out, f, indent=0)
    e = np.array([[abs(w["e_test_fp64"] - w["e_test_quad"]) / w["e_test_quad"], w["lam"]] for w in rows])
    print("%s: %d trials, status %d; fp64-vs-quad test-energy deviation: max %.2e (lambda %.1e), median %.2e" %
          (name, len(rows), r["status"], e[:, 0].max(), e[e[:, 0].argmax(), 1], np.median(e[:, 0])), flush=True)


# Free runs in quad precision to the reference's own stop: the trajectory exact arithmetic follows from the same
# start.  Two fp64 solvers part from it (and from each other) after ~20 trials -- the problem amplifies rounding by ~10x per
# iteration -- so "final cost to 1e-6" between two fp64 sides is not decidable; what is: whose final energy lies closer to the
# quad run's.  One run is sequential, ~4 CPU-seconds per trial (10 - 20 minutes).
FREE_RUNS = {
    "freerun_problem21_qrchol": (("bal", "problem-21-11315-pre.txt"), O.QRCHOL),
    "freerun_problem21_cholesky": (("bal", "problem-21-11315-pre.txt"), O.CHOLESKY),
}


def free_run(name):
    source, kind = FREE_RUNS[name]
    p = load(source)
    r = O.referee_minimize(kind, p, max_trials=2000)
    tr = r["trace"]
    _, e_final = O.residuals(p, r["cam15"], r["pts"])
    o = O.minimize(kind, p)  # the fp64 oracle's own free run, for the record
    out = dict(case=name, source=[source[0], source[1]], kind=int(kind), N=p.N, M=p.M, K=p.K, status=int(r["status"]), trials=int(len(tr)),
               final_energy_quad=float(O.referee_energy(p, r["cam15"], r["pts"])), final_energy_of_rounded_state_fp64=float(e_final),
               last_row_energy_quad=float(tr[-1][6] if tr[-1][1] else tr[-1][2]),
               oracle_fp64=dict(status=int(o["status"]), trials=int(len(o["trace"])),
                                final_energy=float(O.residuals(p, o["cam15"], o["pts"])[1])),
               note="quad: oracle/ba_referee.c ref_minimize (ora_minimize in __float128) from the file's start; the energies are of the "
                    "state the loop leaves behind (the flat-line exit happens before x = xTest, BacktrackLevMarqQRChol.h:419-428)",
               trace=[[int(w[0]), int(w[1]), float(w[2]), float(w[3]), float(w[4])] for w in tr])
    with open(os.path.join(HERE, "referee_%s.json" % name), "w") as f:
        json.dump(out, f, indent=0)
    print("%s: quad run %d trials, status %d, final energy %.9g; fp64 oracle %d trials, status %d, final energy %.9g" %
          (name, len(tr), r["status"], out["final_energy_quad"], len(o["trace"]), o["status"], out["oracle_fp64"]["final_energy"]), flush=True)


# What "the final cost" of this algorithm is worth as a number: the fp64 oracle's own free run from inputs perturbed by 1e-13
# (relative, seeded) -- far below anything the BAL files' 17 printed digits mean.  The run amplifies that by ~10x per iteration, so
# the runs part after ~15 iterations and stop (same flat-line test) at different energies: the spread of the ensemble is the
# resolution at which ANY implementation's final energy can be compared with the reference's.
ENSEMBLES = {
    # (round 4: 64 members each -- members 0..15 are round 3's -- so that a two-sample rank test against as many GPU runs of the
    # same perturbed inputs has the power to see a shift of a fraction of the spread; and config 1's stand-in)
    "ensemble_problem21_qrchol": (("bal", "problem-21-11315-pre.txt"), O.QRCHOL, 64),
    "ensemble_problem21_cholesky": (("bal", "problem-21-11315-pre.txt"), O.CHOLESKY, 64),
    "ensemble_cfg1_cholesky": (("synthetic", (16, 22106, 83718, 1001)), O.CHOLESKY, 64),
    # the same ensembles with the oracle instantiated in long double (x87 extended, eps 1.1e-19; oracle/ba_referee.c: ref80_minimize):
    # the reference algorithm with 2048x less rounding noise in every operation.  The distribution of its final energies is NOT the
    # fp64 oracle's (problem-21 QRCHOL: median 1462.4 against 1455.0) -- where this algorithm stops depends on how noisy its linear
    # solves are at the lambda floor, and an implementation can only be compared with the ensemble of matching noise.
    # ... and with ONLY the long sums of the reduced camera system (S, rhs, g_c: thousands of terms per entry) accumulated in long double,
    # everything else the fp64 oracle as it is (ora_set_wide_sums): this alone moves the fp64 ensemble onto the x87 one -- the noise
    # that decides where the free run stops is the one-after-the-other summation of S.  The GPU sums S in fixed trees / MFMA chunks.
    "ensemble_problem21_qrchol_widesums": (("bal", "problem-21-11315-pre.txt"), O.QRCHOL, 64, "widesums"),
    "ensemble_problem21_cholesky_widesums": (("bal", "problem-21-11315-pre.txt"), O.CHOLESKY, 64, "widesums"),
    "ensemble_cfg1_cholesky_widesums": (("synthetic", (16, 22106, 83718, 1001)), O.CHOLESKY, 64, "widesums"),
    "ensemble_problem21_qrchol_x87": (("bal", "problem-21-11315-pre.txt"), O.QRCHOL, 64, "x87"),
    "ensemble_problem21_cholesky_x87": (("bal", "problem-21-11315-pre.txt"), O.CHOLESKY, 64, "x87"),
    "ensemble_cfg1_cholesky_x87": (("synthetic", (16, 22106, 83718, 1001)), O.CHOLESKY, 64, "x87"),
}


def _member(args):
    name, k = args
    source, kind = ENSEMBLES[name][:2]
    p = O.ensemble_member(load(source), k)  # member 0 is the unperturbed input
    mode = ENSEMBLES[name][3] if len(ENSEMBLES[name]) > 3 else "fp64"
    O.set_wide_sums(mode == "widesums")
    r = O.referee_minimize(kind, p, max_trials=5000, x87=True) if mode == "x87" else O.minimize(kind, p)
    O.set_wide_sums(False)
    return k, int(r["status"]), int(len(r["trace"])), float(O.residuals(p, r["cam15"], r["pts"])[1])


def ensemble(name, pool):
    source, kind, n = ENSEMBLES[name][:3]
    res = sorted(pool.map(_member, [(name, k) for k in range(n)]))
    e = np.array([w[3] for w in res])
    out = dict(case=name, source=[source[0], source[1] if isinstance(source[1], str) else list(source[1])], kind=int(kind), perturbation=1e-13,
               members=[dict(member=w[0], status=w[1], trials=w[2], final_energy=w[3]) for w in res],
               final_energy_min=float(e.min()), final_energy_max=float(e.max()), final_energy_median=float(np.median(e)),
               arithmetic={"x87": "x87 long double", "widesums": "fp64, S / rhs / g_c accumulated in long double", "fp64": "fp64"}[
                   ENSEMBLES[name][3] if len(ENSEMBLES[name]) > 3 else "fp64"],
               note="oracle free runs (oracle/ba_oracle_impl.h) to the reference's own stop from inputs perturbed by 1e-13 relative; member 0 "
                    "unperturbed; final energies by the fp64 residual function of the final state rounded to double")
    with open(os.path.join(HERE, "referee_%s.json" % name), "w") as f:
        json.dump(out, f, indent=0)
    print("%s: final energies %.6g .. %.6g (median %.6g), trials %d .. %d" % (name, e.min(), e.max(), np.median(e), min(w[2] for w in res), max(w[2] for w in res)),
          flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    en = [n for n in names if n in ENSEMBLES]
    names = [n for n in names if n not in ENSEMBLES]
    if en:
        with Pool(int(os.environ.get("REFEREE_PROCS", "8"))) as pool:
            for n in en:
                ensemble(n, pool)
    fr = [n for n in names if n in FREE_RUNS]
    names = [n for n in names if n not in FREE_RUNS]
    if names:
        with Pool(int(os.environ.get("REFEREE_PROCS", "8"))) as pool:
            for n in names:
                make(n, pool)
    if fr:
        with Pool(len(fr)) as pool:
            pool.map(free_run, fr)
