"""GPU vs the quad-precision referee, per trial, along the fp64 oracle's whole LM trajectory.

The reference ships no golden output and cannot be built offline, so the fp64 oracle is "parity unpinned", and where lambda sits
at its 1e-10 floor (cond(J'J + lambda I) ~ 1e20) any two fp64 solvers disagree about a step by far more than 1e-6.  What CAN be
decided there is who is closer to the truth.  tests/golden/referee_*.json (tests/golden/make_referee.py, oracle/ba_referee.c) holds,
for every trial of the oracle's free run to the reference's own stop, the trial evaluated in __float128 from the oracle's state
(x_k, lambda_k).  Here the same state is injected into the GPU solver and its test energy is measured against the quad value,
next to the oracle's own error:

  * the oracle replay must be on the fixture's trajectory (its energies are bit-sensitive fingerprints of x_k);
  * the energy at x_k (no linear solve involved) agrees with quad to 1e-13;
  * per trial |e_gpu - e_quad| <= max(4 |e_oracle - e_quad|, 4 x the largest oracle error among the trials within one decade of
    lambda either side) + 1e-13: the GPU is never an outlier of its regime -- no lambda is exempt (round 1 made no claim below
    lambda = 1e-9);
  * per lambda decade with at least four trials the median GPU error is at most 3 x the median oracle error, and over all trials
    the geometric mean of (GPU error / oracle error) is at most 2: as close to the truth as the oracle (measured: 0.2 - 0.8);
  * where the quad step decides accept / reject by a margin beyond both fp64 errors, the GPU decides the same.
"""
import json
import os

import numpy as np
import pytest

from conftest import DATA21, DATA39, ROOT

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = ["problem21_qrchol", "problem21_cholesky", "problem21_moreqr", "problem39_qrchol", "synthetic60_cholesky", "cfg1_cholesky",
         "cfg4_cholesky"]


def load_case(ba, O, name):
    with open(os.path.join(GOLD, "referee_%s.json" % name)) as f:
        fx = json.load(f)
    if fx["source"][0] == "bal":
        pg = ba.Problem.load_bal(os.path.join(ROOT, "data", fx["source"][1]))
    else:
        pg = ba.Problem.synthetic(*fx["source"][1])
    a = pg.arrays()
    po = O.Problem(pg.N, pg.M, pg.K, a["cam_idx"], a["pt_idx"], a["meas"], a["cams9"], a["pts"])
    return fx, pg, po


def measure(ba, O, name):
    """-> rows of (lambda, E_fp64, err_oracle, err_gpu, e_quad, e_gpu, E_gpu_err, accepted_quad, margin) per trial."""
    fx, pg, po = load_case(ba, O, name)
    tr = fx["trials"]
    r = O.minimize(fx["kind"], po, max_trials=len(tr), snapshots=True)
    assert len(r["trace"]) == len(tr)
    e_replay = r["trace"][:, 2]
    e_fix = np.array([w["energy_fp64"] for w in tr])
    # the replayed oracle must BE the fixture's trajectory, or the quad columns describe other states
    assert np.allclose(e_replay, e_fix, rtol=1e-14, atol=0), "oracle replay left the fixture's trajectory at trial %d" % int(
        np.argmax(np.abs(e_replay - e_fix) > 1e-14 * e_fix))
    s = ba.Solver(pg, fx["kind"], ba.F64)
    rows = []
    N = po.N
    for k, w in enumerate(tr):
        x = r["snap"][k]
        s.set_state(x[: 15 * N].reshape(N, 15), x[15 * N:])
        e, _ = s.linearize(False)
        et, rs, dn = s.try_step(w["lam"])
        eq = w["e_test_quad"]
        rows.append((w["lam"], w["energy_fp64"], abs(w["e_test_fp64"] - eq) / eq, abs(et - eq) / eq, eq, et,
                     abs(e - w["energy_quad"]) / w["energy_quad"], float(eq < w["energy_quad"]),
                     abs(eq - w["energy_quad"]) / w["energy_quad"]))
    return np.array(rows)


@pytest.mark.parametrize("name", CASES)
def test_gpu_is_as_close_to_quad_as_the_oracle(ba, O, gpu_ok, name):
    m = measure(ba, O, name)
    lam, err_o, err_g = m[:, 0], m[:, 2], m[:, 3]
    assert np.all(np.isfinite(m[:, 5]))
    assert m[:, 6].max() < 1e-13  # energy at x_k: evaluation only
    dec = np.floor(np.log10(lam)).astype(int)
    table = [(d, int((dec == d).sum()), np.median(err_o[dec == d]), np.median(err_g[dec == d]), err_o[dec == d].max(), err_g[dec == d].max())
             for d in np.unique(dec)]
    print("\n%s: %d trials; per lambda decade: n, median / max error vs quad of the fp64 oracle and of the GPU" % (name, len(m)))
    for d, n, mo, mg, xo, xg in table:
        print("   1e%+03d  n=%3d  oracle %.2e / %.2e   gpu %.2e / %.2e" % (d, n, mo, xo, mg, xg))
    ll = np.log10(lam)
    for k in range(len(m)):  # never an outlier of its regime: the oracle errors within one decade of lambda either side set the cap
        cap = 4 * err_o[np.abs(ll - ll[k]) <= 1.0].max()
        assert err_g[k] <= max(4 * err_o[k], cap) + 1e-13, (name, k, lam[k], err_g[k], err_o[k], cap)
    for d, n, mo, mg, xo, xg in table:  # as close to the truth as the oracle, wherever a decade holds enough trials for a median
        if n >= 4:
            assert mg <= 3 * mo + 1e-13, (name, "lambda decade 1e%d" % d, mg, mo)
            # ... and an ABSOLUTE guard (the criteria above are relative to the oracle's own error, which reaches
            # 0.5 in single trials at the lambda floor): whatever the oracle does, the GPU's median error against quad in any decade
            # stays below 2e-4 (measured at the 1e-10 floor: 5e-11 ... 4e-5 over the seven cases; 1e-5 and less above it)
            assert mg <= 2e-4, (name, "lambda decade 1e%d" % d, mg)
    ratio = np.maximum(err_g, 1e-15) / np.maximum(err_o, 1e-15)
    gm = float(np.exp(np.mean(np.log(ratio))))
    print("   geometric mean of (GPU error / oracle error) over all trials: %.2f; GPU closer to quad in %d of %d trials" %
          (gm, int((err_g < err_o).sum()), len(m)))
    assert gm <= 2.0, (name, gm)
    # accept / reject against the truth wherever the truth is decisive for both fp64 sides
    decisive = m[:, 8] > 10 * np.maximum(err_o, err_g) + 1e-12
    acc_g = m[:, 5] < m[:, 1]
    assert np.array_equal(acc_g[decisive], m[decisive, 7] > 0.5)


# ---- fp32 (config 3: QRKIT, Scalar = float) --------------------------------------------------------------------------------------------
# round 2 asserted config 3 at trial 0 only.  tests/golden/referee_problem39_qrkit_f32.json holds the first 24 trials
# of the FP32 oracle's own free run on problem-39 under QRKIT (dense Householder QR of J2bot in float), each evaluated in quad from the
# same (float) state; ..._states.npz holds the states.  The GPU (BA_QRKIT, BA_F32: k_elim_qr + the dense QR of ba_qr.hip.h) gets the same
# state and lambda per trial.  Same assertions as the fp64 cases, with the floor of a float evaluation (1e-6) in place of 1e-13.
# Then the GPU's OWN trajectory: a Levenberg-Marquardt run driven through the step-level seam (GPU steps, the reference's accept / lambda
# rule), and at every third of its first 24 states the quad value and the fp32 oracle's step from that very state.

def _fp32_sides(args):
    """(state, lambda) -> (fp32 oracle's test energy, quad test energy): CPU, 18 + 8 seconds."""
    import oracle_lib as OL
    x, lam = args
    p = OL.load_bal(DATA39)
    N = p.N
    cam = x[: 15 * N].astype(np.float32)
    pts = x[15 * N:].astype(np.float32)
    f, _ = OL.residuals(p, cam, pts)
    Jc, Jp = OL.jacobian(p, cam, pts)
    st = OL.step(OL.QRKIT, p, Jc, Jp, f, np.float32(lam), want_S=False)
    co, pt = OL.retract(p, cam, pts, st["dx"])
    _, eo = OL.residuals(p, co, pt)
    q = OL.referee_trial(OL.QRCHOL, p, x[: 15 * N], x[15 * N:], float(lam))
    return float(eo), float(q["e_test"]), float(q["energy"])


def _regime_asserts(name, lam, err_o, err_g, floor):
    ll = np.log10(lam)
    for k in range(len(lam)):
        cap = 4 * err_o[np.abs(ll - ll[k]) <= 1.0].max()
        assert err_g[k] <= max(4 * err_o[k], cap) + floor, (name, k, lam[k], err_g[k], err_o[k], cap)
    ratio = np.maximum(err_g, floor) / np.maximum(err_o, floor)
    gm = float(np.exp(np.mean(np.log(ratio))))
    print("   %s: %d trials, geometric mean of (GPU error / fp32-oracle error) %.2f; GPU closer to quad in %d; medians oracle %.2e gpu %.2e" %
          (name, len(lam), gm, int((err_g < err_o).sum()), np.median(err_o), np.median(err_g)))
    assert gm <= 2.0, (name, gm)
    assert np.median(err_g) <= 3 * np.median(err_o) + floor, (name, np.median(err_g), np.median(err_o))


@pytest.mark.timeout(900)
def test_fp32_qrkit_is_as_close_to_quad_as_the_fp32_oracle(ba, O, gpu_ok):
    with open(os.path.join(GOLD, "referee_problem39_qrkit_f32.json")) as f:
        fx = json.load(f)
    z = np.load(os.path.join(GOLD, "referee_problem39_qrkit_f32_states.npz"))
    pg = ba.Problem.load_bal(DATA39)
    N = pg.N
    s = ba.Solver(pg, ba.QRKIT, ba.F32)
    tr = fx["trials"]
    rows = []
    for k, w in enumerate(tr):
        x = z["states"][z["state_of_trial"][k]].astype(np.float64)
        s.set_state(x[: 15 * N].reshape(N, 15), x[15 * N:])
        e, _ = s.linearize(False)
        et, _, _ = s.try_step(w["lam"])
        eq = w["e_test_quad"]
        rows.append((w["lam"], abs(w["e_test_fp64"] - eq) / eq, abs(et - eq) / eq, abs(e - w["energy_quad"]) / w["energy_quad"], et, w["energy_fp64"], eq, w["energy_quad"]))
    m = np.array(rows)
    assert np.all(np.isfinite(m[:, 4]))
    assert m[:, 3].max() < 5e-6  # energy at x_k in float: evaluation only
    print("\nproblem39 QRKIT fp32, the fp32 oracle's trajectory:")
    _regime_asserts("injected", m[:, 0], m[:, 1], m[:, 2], 1e-6)
    # accept / reject against the truth wherever the truth is decisive for both float sides
    margin = np.abs(m[:, 6] - m[:, 7]) / m[:, 7]
    decisive = margin > 10 * np.maximum(m[:, 1], m[:, 2]) + 1e-5
    assert np.array_equal((m[:, 4] < m[:, 5])[decisive], (m[:, 6] < m[:, 7])[decisive])

    # ---- the GPU's own trajectory
    x0 = z["states"][0].astype(np.float64)
    s.set_state(x0[: 15 * N].reshape(N, 15), x0[15 * N:])
    e, dmax = s.linearize()
    lam, inc = float(np.float32(1e-12 * dmax)), 2.0
    own = []
    for k in range(24):
        x = np.concatenate([s.get(ba.GET_CAMS), s.get(ba.GET_POINTS)])
        et, rs, _ = s.try_step(lam)
        own.append((x, lam, et))
        if et < e:  # BacktrackLevMarqQRChol.h:374-394
            rho = (e - et) / rs
            lam = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), 1e-10)
            inc = 2.0
            s.accept()
            e, _ = s.linearize(False)
        else:       # :395-410
            lam *= inc
            inc = inc ** 1.5
        lam = float(np.float32(lam))
    # (a float trajectory can run into a state whose Jacobian is not finite -- an observation's squared error overflows once its point has
    # drifted onto the camera's plane; every later trial is then NaN = rejected, like in the reference's own float arithmetic, DESIGN.md
    # section 2: the comparison uses the trials in front of that)
    finite = [w for w in own if np.isfinite(w[2])]
    assert len(finite) >= 6, len(finite)
    pick = finite[::3] if len(finite) >= 18 else finite[::2]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(8, len(pick))) as pool:
        sides = pool.map(_fp32_sides, [(x, lam) for x, lam, _ in pick])
    lam = np.array([w[1] for w in pick])
    eg = np.array([w[2] for w in pick])
    eo = np.array([w[0] for w in sides])
    eq = np.array([w[1] for w in sides])
    print("problem39 QRKIT fp32, the GPU's own trajectory (every third of its first 24 states):")
    _regime_asserts("own", lam, np.abs(eo - eq) / eq, np.abs(eg - eq) / eq, 1e-6)
