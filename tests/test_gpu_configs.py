"""GPU parity on the BASELINE.json configurations themselves ("configs untested").

  cfg4  CHOLESKY, problem-257-65132 (file missing from the reference checkout: seeded stand-in of the same N, M, K), fp64, full size
  cfg5  QRCHOL, synthetic 1024 cameras x 500 k points x 4 M observations, fp64, full size
  cfg3  QRKIT symbol, problem-39-18060-pre.txt, Scalar = float
  cfg1  CHOLESKY, problem-16-22106 stand-in, fp64 (the "CPU plumbing" config, here at its dimensions on the GPU)
and the production LM loop (hipGraph replay, device-side step control) across REJECTED trials, plus the executables' stdout
protocol.  Tolerances as in test_gpu_parity.py: residual / Jacobian 1e-11 of the array maximum, energy 1e-12, S 1e-11 of max|S|,
step 1e-6, backward error of the step in the normal equations 1e-9.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import DATA21, ROOT, to_oracle

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def backward_error(po, Jc, Jp, f, dx, lam, want_g=False):
    """|(J'J + lam I) dx + J'r| / |J'r| with the oracle's Jacobian (and g = -J'r on request)."""
    M, N = po.M, po.N
    Jdx = np.einsum("krc,kc->kr", Jc, dx[3 * M:].reshape(N, 9)[po.cam_idx]) + \
        np.einsum("krc,kc->kr", Jp, dx[:3 * M].reshape(M, 3)[po.pt_idx])
    JtJdx = np.zeros_like(dx)
    g = np.zeros_like(dx)
    fr = f.reshape(-1, 2)
    for out, v in ((JtJdx, Jdx), (g, -fr)):
        np.add.at(out[3 * M:].reshape(N, 9), po.cam_idx, np.einsum("krc,kr->kc", Jc, v))
        np.add.at(out[:3 * M].reshape(M, 3), po.pt_idx, np.einsum("krc,kr->kc", Jp, v))
    be = np.linalg.norm(JtJdx + lam * dx - g) / np.linalg.norm(g)
    return (be, g) if want_g else be


# ---- cfg4 -------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def cfg4(ba):
    return ba.Problem.synthetic(257, 65132, 225911, 1004)


def test_cfg4_linearize_and_step_match_oracle(ba, O, gpu_ok, cfg4):
    po = to_oracle(cfg4)
    cam = O.init_cams(po)
    f, e = O.residuals(po, cam, po.pts)
    Jc, Jp = O.jacobian(po, cam, po.pts)
    s = ba.Solver(cfg4, ba.CHOLESKY, ba.F64)
    s.keep_intermediates(True)
    eg, dmax = s.linearize()
    assert abs(eg - e) <= 1e-12 * e
    assert relmax(s.get(ba.GET_RESIDUALS), f) < 1e-11
    assert relmax(s.get(ba.GET_JC).reshape(-1, 2, 9), Jc) < 1e-11
    assert relmax(s.get(ba.GET_JP).reshape(-1, 2, 3), Jp) < 1e-11
    lam = 1e-12 * dmax  # the symbol's own lambda0
    st = O.step(O.CHOLESKY, po, Jc, Jp, f, lam)
    assert abs(dmax - st["diagmax"]) <= 1e-12 * dmax
    assert relmax(s.get(ba.GET_GRAD), st["g"]) < 1e-11
    et, rs, dn = s.try_step(lam)
    # Reduced camera matrix.  GPU and fp64 oracle differ by 3.6e-11 of max|S| here, beyond the 1e-11 the small problems show: the
    # stand-in's cameras sit 0.24 apart on their ring, so two-view points have near-singular 3x3 blocks that lambda0 = 3.6e-5 barely
    # regularises, and the two fp64 eliminations (FMA contraction or not) differ in them.  Neither is the yardstick: the same
    # assembly in __float128 is (oracle/ba_referee.c, ~1 CPU-minute) -- the GPU must be within 1e-11, or as close to it as the oracle.
    Sg, Sq = s.get(ba.GET_S), None
    Sq, rq = O.referee_reduced(O.CHOLESKY, po, cam, po.pts, lam)
    err_g, err_o = relmax(Sg, Sq), relmax(st["S"], Sq)
    print("cfg4 reduced matrix vs quad: gpu %.2e, fp64 oracle %.2e (of max|S|); gpu vs oracle %.2e" % (err_g, err_o, relmax(Sg, st["S"])))
    assert err_g < max(1e-11, 2 * err_o)
    assert relmax(Sg, st["S"]) < 1e-10
    assert relmax(s.get(ba.GET_RHS), rq) < max(1e-10, 2 * relmax(st["rhs"], rq))
    dx = s.get(ba.GET_DX)
    assert np.linalg.norm(dx - st["dx"]) < 1e-6 * np.linalg.norm(st["dx"])
    assert backward_error(po, Jc, Jp, f, dx, lam) < 1e-9
    co, pt = O.retract(po, cam, po.pts, st["dx"])
    _, e_or = O.residuals(po, co, pt)
    assert abs(et - e_or) < 1e-7 * e_or


def test_cfg4_free_running_prefix(ba, O, gpu_ok, cfg4):
    """The production loop (graph replay) on the headline workload: 5 table rows against the oracle -- the same accept / reject
    sequence, energies to 1e-7 over the first three rows.  This problem starts at cond(J'J + lambda I) = 1e12 (lambda0 = 3.6e-5
    against max diag 3.6e7) and lambda falls by 3 per row: by row 4 the two fp64 solvers are 1e-6 apart and by row 5 2e-4
    (measured; on problem-21 that point comes at row 8).  Which of the two is closer to the exact step there is decided per trial
    against __float128 in test_gpu_referee.py (case cfg4_cholesky), not here."""
    po = to_oracle(cfg4)
    ro = O.minimize(O.CHOLESKY, po, max_trials=5)
    s = ba.Solver(cfg4, ba.CHOLESKY, ba.F64)
    rg = s.minimize(max_trials=5)
    tg, to = rg["trace"], ro["trace"]
    assert tg.shape[0] == to.shape[0] == 5
    assert np.array_equal(tg[:, 0], to[:, 0]) and np.array_equal(tg[:, 1], to[:, 1])
    assert np.allclose(tg[:3, 2], to[:3, 2], rtol=1e-7)
    assert np.allclose(tg[:, 2], to[:, 2], rtol=1e-2)  # still the same descent
    assert np.allclose(tg[:3, 3], to[:3, 3], rtol=1e-4) and np.allclose(tg[:3, 4], to[:3, 4], rtol=1e-4)


# ---- cfg1 (at its dimensions) -------------------------------------------------------------------------------------

def test_cfg1_dimensions_step_and_prefix(ba, O, gpu_ok):
    p = ba.Problem.synthetic(16, 22106, 83718, 1001)
    po = to_oracle(p)
    cam = O.init_cams(po)
    f, e = O.residuals(po, cam, po.pts)
    Jc, Jp = O.jacobian(po, cam, po.pts)
    s = ba.Solver(p, ba.CHOLESKY, ba.F64)
    s.keep_intermediates(True)
    eg, dmax = s.linearize()
    assert abs(eg - e) <= 1e-12 * e
    lam = 1e-12 * dmax
    st = O.step(O.CHOLESKY, po, Jc, Jp, f, lam)
    s.try_step(lam)
    assert relmax(s.get(ba.GET_S), st["S"]) < 1e-11
    dx = s.get(ba.GET_DX)
    assert np.linalg.norm(dx - st["dx"]) < 1e-6 * np.linalg.norm(st["dx"])
    ro = O.minimize(O.CHOLESKY, po, max_trials=6)
    s2 = ba.Solver(p, ba.CHOLESKY, ba.F64)
    rg = s2.minimize(max_trials=6)
    assert np.array_equal(rg["trace"][:, 1], ro["trace"][:, 1])
    assert np.allclose(rg["trace"][:4, 2], ro["trace"][:4, 2], rtol=1e-7)  # (row 5 is 2e-4 apart: see test_cfg4_free_running_prefix)
    assert np.allclose(rg["trace"][:, 2], ro["trace"][:, 2], rtol=1e-2)  # still the same descent


# ---- cfg5 -------------------------------------------------------------------------------------------------------

def test_cfg5_full_size(ba, O, gpu_ok):
    """D = 9216, 4 M observations.  The plain-C oracle's dense LDL^T at this size takes minutes, so:
      * residuals, Jacobian, energy, gradient of the FULL problem against the oracle (cheap, O(K));
      * the reduced camera matrix S and its rhs against the oracle's elimination + assembly (no factorisation) on a point
        subset of the same problem (first 20 000 points, all 1024 cameras -> the same 9216 x 9216 system layout);
      * the full step by its backward error in the normal equations and by the energy decrease it produces.
    The two-workgroups-per-CU dense-factor variant with its pair phase (144 block columns: macro-tile updates by pairs of panels
    for the first 72 steps, single-panel steps behind) and the data-flow back sweep with 72 groups are the ones that run here."""
    p = ba.Problem.synthetic(1024, 500000, 4000000, 1005)
    po = to_oracle(p)
    cam = O.init_cams(po)
    f, e = O.residuals(po, cam, po.pts)
    Jc, Jp = O.jacobian(po, cam, po.pts)
    s = ba.Solver(p, ba.QRCHOL, ba.F64)
    eg, dmax = s.linearize()
    assert abs(eg - e) <= 1e-12 * e
    assert relmax(s.get(ba.GET_RESIDUALS), f) < 1e-11
    jc = s.get(ba.GET_JC).reshape(-1, 2, 9)
    assert relmax(jc, Jc) < 1e-11
    del jc
    assert relmax(s.get(ba.GET_JP).reshape(-1, 2, 3), Jp) < 1e-11
    lam = 1e-12 * dmax
    et, rs, dn = s.try_step(lam)
    assert np.isfinite(et) and et < e
    dx = s.get(ba.GET_DX)
    be, g = backward_error(po, Jc, Jp, f, dx, lam, want_g=True)
    assert be < 1e-9
    assert relmax(s.get(ba.GET_GRAD), g) < 1e-11
    co, pt = O.retract(po, cam, po.pts, dx)
    _, e_or = O.residuals(po, co, pt)
    assert abs(et - e_or) < 1e-10 * e_or  # the test energy at the GPU's own trial point, evaluated by the oracle
    rho_den = float(dx @ (lam * dx + g))
    assert abs(rs - rho_den) < 1e-9 * abs(rho_den) and abs(dn - np.linalg.norm(dx)) < 1e-10 * dn
    del s, Jc, Jp, f, dx
    # assembly parity on a subset (same cameras, first 20 000 points)
    a = p.arrays()
    npts = 20000
    k = int(np.searchsorted(a["pt_idx"], npts, side="left"))
    ps = ba.Problem.from_arrays(p.N, npts, k, a["cam_idx"][:k], a["pt_idx"][:k], a["meas"][:2 * k], a["cams9"], a["pts"][:3 * npts])
    pso = to_oracle(ps)
    cs = O.init_cams(pso)
    fs, es = O.residuals(pso, cs, pso.pts)
    Jcs, Jps = O.jacobian(pso, cs, pso.pts)
    st = O.step(O.QRCHOL | O.ASSEMBLE_ONLY, pso, Jcs, Jps, fs, lam)
    s2 = ba.Solver(ps, ba.QRCHOL, ba.F64)
    s2.keep_intermediates(True)
    s2.linearize()
    s2.try_step(lam)
    assert relmax(s2.get(ba.GET_S), st["S"]) < 1e-11
    assert relmax(s2.get(ba.GET_RHS), st["rhs"]) < 1e-10


# ---- cfg3 -------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("kind", [0, 1])
def test_cfg3_f32_lm(ba, O, gpu_ok, prob39, kind):
    """Scalar = float on problem-39 under the QRKIT symbol (config 3: per-point QR + dense Householder QR of J2bot, 181 k x 351)
    and under QRCHOL: the accepted energies decrease, the first trial agrees with the fp32 oracle of the same symbol (energy
    before 1e-5, accept decision, test energy 2e-3 -- or, when two fp32 solvers differ by more than that, the GPU is at least
    as close as the fp32 oracle to the quad value of that trial, tests/golden/referee_problem39_qrchol.json trial 0; the
    oracle's dense QR of that matrix takes ~1/2 minute per trial), and the robust statistics after the run are finite."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "referee_problem39_qrchol.json")) as f:
        quad = json.load(f)["trials"][0]  # QRKIT and QRCHOL solve the same system from the same start and lambda0
    po = to_oracle(prob39)
    ro = O.minimize(kind, po, dtype=np.float32, max_trials=2)["trace"]
    s = ba.Solver(prob39, kind, ba.F32)
    r = s.minimize(max_trials=15)
    tg = r["trace"]
    acc = tg[tg[:, 1] == 1]
    assert len(acc) >= 3 and np.all(np.diff(acc[:, 2]) < 0)
    assert abs(tg[0, 2] - ro[0, 2]) < 1e-5 * ro[0, 2]
    assert tg[0, 1] == ro[0, 1] == 1
    assert abs(quad["lam"] - ro[0, 5]) < 1e-4 * quad["lam"] and abs(quad["energy_quad"] - tg[0, 2]) < 1e-5 * tg[0, 2]  # same trial
    eq = quad["e_test_quad"]  # f of row 1 = test energy of the first accepted step
    assert abs(tg[1, 2] - ro[1, 2]) < 2e-3 * ro[1, 2] or abs(tg[1, 2] - eq) <= abs(ro[1, 2] - eq), (tg[1, 2], ro[1, 2], eq)
    st = s.stats()
    # (the plain mean of the errors may overflow float: a few outlier points drift to depth ~ 0 and their raw error exceeds 1e19, on
    # the GPU and in the fp32 oracle alike -- profiles/r02_configs_to_termination.jsonl; the robust figures stay finite)
    assert all(np.isfinite(st[k]) for k in ("inlier_mean_err", "objective")) and not np.isnan(st["mean_err"])


def test_cfg3_reflector_of_a_column_with_a_denormal_norm(ba, gpu_ok, prob39, monkeypatch):
    """round 3 saw config 3 accept no LM step when the reflectors' beta came from the bare v_sqrt_f32, and
    stepped around it.  The cause (round 4, scripts/diag_hw_sqrt.py): problem-39's k1 columns hold entries of ~1e-21 in some 1024-row
    chunks; their SQUARED norm is a denormal float, which sqrtf maps to a normal number and v_sqrt_f32 flushes to zero -- beta = 0,
    tau = 0 * inf = NaN, and NaN in every later column of the panel (first: column 70).  Not a knife edge of the algorithm: a missing
    guard.  k_qr_chunk now forms the norm of such a column from rescaled entries (and treats sqrt(...) == 0 like a zero column), so the
    square root's flavour no longer matters: the default (v_sqrt_f32 + one Newton step), BA_QR_HW_SQRT=1 (the bare instruction) and
    BA_QR_HW_SQRT=2 (IEEE sqrtf, rounds 2 - 3's) must give the same first step to float precision and the same accept decisions."""
    ref = ba.Solver(prob39, ba.QRKIT, ba.F32)
    e0, dmax = ref.linearize()
    et0, _, dn0 = ref.try_step(1e-12 * dmax)
    r0 = ref.minimize(max_trials=6)["trace"]
    del ref
    try:
        for flavour in ("1", "2"):
            monkeypatch.setenv("BA_QR_HW_SQRT", flavour)
            s = ba.Solver(prob39, ba.QRKIT, ba.F32)
            e1, _ = s.linearize()
            et1, _, dn1 = s.try_step(1e-12 * dmax)
            r1 = s.minimize(max_trials=6)["trace"]
            del s
            assert np.isfinite(et1) and np.isfinite(dn1) and e1 == e0
            assert abs(et1 - et0) < 1e-4 * et0 and abs(dn1 - dn0) < 1e-3 * dn0, (flavour, et0, et1, dn0, dn1)
            assert np.array_equal(r1[:, 1], r0[:, 1]) and r1[:, 1].sum() >= 4, flavour
    finally:
        monkeypatch.setenv("BA_QR_HW_SQRT", "0")
        ba.Solver(prob39, ba.QRKIT, ba.F32)  # (the switch is a device-side flag of the library, written at solver creation: back to the default)


# ---- production loop across rejected trials ------------------------------------------------------------------------

@pytest.mark.parametrize("kind", [2, 1, 3])
def test_minimize_through_rejections(ba, O, gpu_ok, kind):
    """synthetic(5, 60, 200, 23): the oracle rejects the first four trials (lambda0 = 2.3e-6 ... 6e-5, by margins of 2-4 %),
    accepts the fifth and then descends.  ba_minimize -- the graph-replayed production loop, not try_step -- must show the same
    accept / reject sequence, the lambda schedule of the retries (x 2, x 2^1.5, ..., BacktrackLevMarqQRChol.h:408-409) and the
    same energies through and after the rejections."""
    p = ba.Problem.synthetic(5, 60, 200, 23)
    po = to_oracle(p)
    ro = O.minimize(kind, po, max_trials=12)["trace"]
    nrej = int(np.argmax(ro[:, 1] == 1))
    if kind != 3:
        assert nrej == 4  # (the premise of this test; MOREQR starts from another lambda0)
    s = ba.Solver(p, kind, ba.F64)
    rg = s.minimize(max_trials=12)
    tg = rg["trace"]
    assert tg.shape[0] == 12
    assert np.array_equal(tg[:, 0], ro[:, 0]) and np.array_equal(tg[:, 1], ro[:, 1])
    assert np.allclose(tg[:, 4], ro[:, 4], rtol=1e-5)              # lambda column (after the update / before the retry)
    assert np.allclose(tg[:nrej + 6, 2], ro[:nrej + 6, 2], rtol=1e-7)  # f through the rejections and five accepted steps after them
    assert np.allclose(tg[:, 2], ro[:, 2], rtol=1e-4)
    # the state after the run is the oracle's state: the rejected trial points never leaked into x
    rs = O.minimize(kind, po, max_trials=12)
    assert relmax(s.get(ba.GET_CAMS), rs["cam15"]) < 1e-6
    assert relmax(s.get(ba.GET_POINTS), rs["pts"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 1])
def test_folded_launches_change_no_bit(ba, gpu_ok, prob21, kind, monkeypatch):
    """On a single shard k_post_reduce rides on the Schur reduce launch and the trial's three sums on k_lm_control (ba_solver.hip:
    post_folded / ctl_reduces); BA_NO_FOLD=1 (read at solver creation) gives each its own launch again.  Same additions in the same
    order either way: the LM table of a run through rejections and acceptances is equal BIT FOR BIT, and so is the final state."""
    runs = []
    for no_fold in (False, True):
        if no_fold:
            monkeypatch.setenv("BA_NO_FOLD", "1")
        else:
            monkeypatch.delenv("BA_NO_FOLD", raising=False)
        s = ba.Solver(prob21, kind, ba.F64)
        r = s.minimize(max_trials=25)
        runs.append((r["trace"][:, :5].copy(), s.get(ba.GET_CAMS).copy(), s.get(ba.GET_POINTS).copy(), r["energy"]))
        del s
    (ta, ca, pa, ea), (tb, cb, pb, eb) = runs
    assert ta.shape == tb.shape == (25, 5)
    assert np.array_equal(ta, tb) and ea == eb
    assert np.array_equal(ca, cb) and np.array_equal(pa, pb)


# ---- stdout protocol of the executables ---------------------------------------------------------------------------

def _numbers_close(a, b, rtol):
    fa, fb = float(a), float(b)
    return abs(fa - fb) <= rtol * max(abs(fa), abs(fb), 1e-300)


def test_executable_stdout_protocol(ba, O, gpu_ok):
    """bin/Bundle_Adjustment_QRChol data/problem-21-11315-pre.txt with BA_MAX_TRIALS=5 against the text the reference would
    print for the oracle's numbers: bundle_adjustment_large.cpp:61-136 (header, progress lines), Utils.h:39-40,65 (statistics
    before / after), BacktrackLevMarqQRChol.h:65-93 (banner, column header, one row per trial; `f` is the energy BEFORE the step)
    -- every column but Elapsed.  Text must match literally; numbers are printed with 6 significant digits: statistics and the
    f column to 2e-6, rho and lambda (which amplify the 1e-8 energy differences of rows 4-5 by E / (E - E_test) ~ 1e2) to 1e-3."""
    exe = os.path.join(ROOT, "bundleadjustment_benchmarks_amd", "bin", "Bundle_Adjustment_QRChol")
    env = dict(os.environ, BA_MAX_TRIALS="5")
    out = subprocess.run([exe, DATA21], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    po = O.load_bal(DATA21)
    cam = O.init_cams(po)
    st0 = O.stats(po, cam, po.pts)
    r = O.minimize(O.QRCHOL, po, max_trials=5)
    st1 = O.stats(po, r["cam15"], r["pts"])

    def stat_lines(st):
        return ["Mean reprojection error: %g" % st["mean_err"],
                "Inlier mean reprojection error: %g (%d / %d inliers)" % (st["inlier_mean_err"], st["n_inliers"], po.K),
                "True objective: %g" % st["objective"]]
    bar = "-" * 80
    expect = ["N(cameras) = 21, M(points) = 11315, K(measurements) = 36455", "Reading image measurements...", "Done.",
              "Reading cameras params...", "Done.", "Reading 3D points...", "Done."] + stat_lines(st0) + [
        "############################## Backtrack LevMarq ###############################", bar,
        " Iter%15s%15s%15s%15s%15s" % ("Status", "f", "rho", "lambda", "Elapsed"), bar]
    for row in r["trace"]:
        expect.append("%5d%15s%15g%15g%15g" % (row[0], "Accepted" if row[1] else "Rejected", row[2], row[3], row[4]))
    expect += [bar, "lm.minimize(params) ...", "LM finished with status: Running"] + stat_lines(st1)
    assert len(lines) == len(expect), out.stdout
    num = re.compile(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?")
    after = False  # the statistics behind the table: the raw reprojection errors of gross outliers, whose robust energy is flat, move
                   # freely along directions the energy does not see -- 3e-5 between two fp64 solvers after five steps (measured)
    for got, want in zip(lines, expect):
        if want.startswith("lm.minimize"):
            assert re.fullmatch(r"lm\.minimize\(params\) \.\.\. [0-9.e+-]+s", got), got
            after = True
            continue
        row = re.match(r"\s+\d+\s+(Accepted|Rejected)", got) is not None
        if row:
            assert re.fullmatch(r".{65}\s*[0-9.e+-]+s", got), got  # five 15-wide columns, then Elapsed (14 wide + 's')
            got = got[:65]
        # same skeleton once the numbers are masked, same numbers to print precision
        assert num.sub("#", got) == num.sub("#", want), (got, want)
        for q, (a, b) in enumerate(zip(num.findall(got), num.findall(want))):
            assert _numbers_close(a, b, 1e-3 if (row and q >= 2) or after else 2e-6), (got, want)


@pytest.mark.parametrize("case", ["problem21_qrchol", "problem21_cholesky", "cfg1_cholesky"])
def test_final_energy_distribution_matches_the_reference_algorithms(ba, O, gpu_ok, case):
    """north_star asks for the reference's final cost to 1e-6.  On these inputs that number does not exist at 1e-6 for the reference
    algorithm itself: its free run amplifies rounding by ~10x per iteration, and from inputs perturbed by 1e-13 (relative) the fp64 oracle
    stops -- same flat-line test -- anywhere in a band of ~ +-1 % (tests/golden/referee_ensemble_*.json: 64 runs each; even the two
    QUAD-precision free runs, QRCHOL and CHOLESKY, end 4.5e-4 apart: referee_freerun_*.json).  One GPU run against that band cannot
    tell a chance draw from a systematic early stop, so the comparison is between DISTRIBUTIONS: the same 64
    perturbed inputs go through ba_minimize.

    What round 4 found (profiles/r04_final_cost_ensembles.json): the GPU's distribution is NOT the plain fp64
    oracle's -- its median final energy is higher by ~0.4 % (Mann-Whitney p = 1e-13 on problem-21 QRCHOL) at the same number of trials.
    The cause is in the ORACLE: it adds the thousands of terms of every entry of the reduced camera system one after the other, and
    that summation noise, amplified by cond ~ 1e20 at the lambda floor, kicks its free run along the gauge directions (|points| drifts
    by 10 % against the GPU's 1 %) and into lower basins of the robust cost.  The same oracle with ONLY those sums accumulated in long
    double (..._widesums.json), or run entirely in x87 long double (..._x87.json), has the GPU's distribution.  So:
      * every run on either side ends with status Success;
      * the GPU's median final energy and median trial count lie inside the plain fp64 ensemble's [min, max] -- no widening;
      * against the accurately-summed fp64 oracle the two-sample rank test (Mann-Whitney, two-sided) does not tell the GPU apart:
        p > 0.01 on final energy AND on trial count."""
    import ensemble_lib as E
    n = 64
    gpu = E.gpu_members(ba, O, case, n)
    plain = E.compare(E.fixture(case), gpu)
    wide = E.compare(E.fixture(case + "_widesums"), gpu)
    print("\n%s vs the fp64 oracle as it is: %s\n%s vs the fp64 oracle with accurately summed S: %s" % (case, plain, case, wide))
    assert plain["gpu_all_success"] and plain["oracle_all_success"] and wide["oracle_all_success"], (plain, wide)
    assert plain["oracle_energy_min"] <= plain["gpu_energy_median"] <= plain["oracle_energy_max"], plain
    assert plain["oracle_trials_min"] <= plain["gpu_trials_median"] <= plain["oracle_trials_max"], plain
    assert wide["p_energy"] > 0.01 and wide["p_trials"] > 0.01, wide


@pytest.mark.parametrize("kind_name", ["qrchol", "cholesky"])
def test_free_run_prefix_matches_the_quad_free_run(ba, gpu_ok, prob21, kind_name):
    """The tight, deterministic pin beside the ensembles: while the trajectories have not yet bifurcated -- the first eight
    table rows on problem-21 -- the GPU's free run IS the quad-precision free run of the reference algorithm
    (tests/golden/referee_freerun_problem21_*.json, oracle/ba_referee.c in __float128): same accept decisions, energies to 1e-6
    relative (measured: 5e-16 ... 1.3e-7 QRCHOL, ... 6.0e-7 CHOLESKY, whose row 8 is at 1.03e-6; the fp64 oracle itself is at 4e-5 by row 7), lambda to 1e-5."""
    import json
    q = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "referee_freerun_problem21_%s.json" % kind_name)))["trace"])
    kind = {"qrchol": ba.QRCHOL, "cholesky": ba.CHOLESKY}[kind_name]
    g = ba.Solver(prob21, kind, ba.F64).minimize(max_trials=8)["trace"]
    rel = np.abs(g[:8, 2] - q[:8, 2]) / q[:8, 2]
    print("\n%s: |f_gpu - f_quad| / f_quad over the first eight rows: %s" % (kind_name, " ".join("%.1e" % v for v in rel)))
    assert np.array_equal(g[:8, :2], q[:8, :2])
    assert rel.max() <= 1e-6, rel
    assert np.allclose(g[:8, 4], q[:8, 4], rtol=1e-5, atol=0)
