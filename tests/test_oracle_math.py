"""CPU: independent checks of the oracle's mathematics (no reference run is possible):
finite differences for the Jacobian, a scipy sparse solve of the normal equations for the step, algebraic identities
between the three solver symbols, and the quirks of the reference that the restatement keeps."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import to_oracle


@pytest.fixture(scope="module")
def small(ba):
    return to_oracle(ba.Problem.synthetic(8, 300, 1100, 5))


def dense_J(p, Jc, Jp):
    rows, cols, vals = [], [], []
    K, M = p.K, p.M
    for r in range(2):
        for c in range(9):
            rows.append(2 * np.arange(K) + r); cols.append(3 * M + 9 * p.cam_idx + c); vals.append(Jc[:, r, c])
        for c in range(3):
            rows.append(2 * np.arange(K) + r); cols.append(3 * p.pt_idx + c); vals.append(Jp[:, r, c])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(2 * K, 3 * M + 9 * p.N))


def test_jacobian_finite_differences(O, small):
    """Central differences through ora_retract (so the rotation columns are the derivative wrt the left-multiplied
    Rodrigues increment, BAFunctor.h:320-323).  Steps above the 1e-6 Rodrigues cut-off (MathUtils.h:74)."""
    p = small
    cam = O.init_cams(p)
    f0, _ = O.residuals(p, cam, p.pts)
    Jc, Jp = O.jacobian(p, cam, p.pts)
    J = dense_J(p, Jc, Jp).tocsc()
    rng = np.random.default_rng(0)
    n = 3 * p.M + 9 * p.N
    cols = np.concatenate([rng.choice(3 * p.M, 12, replace=False), 3 * p.M + rng.choice(9 * p.N, 27, replace=False)])
    worst = 0.0
    for c in cols:
        kind = (c - 3 * p.M) % 9 if c >= 3 * p.M else -1
        h = {-1: 1e-5, 0: 1e-5, 1: 1e-5, 2: 1e-5, 3: 3e-6, 4: 3e-6, 5: 3e-6, 6: 1e-3, 7: 1e-4, 8: 1e-3}[kind]
        dx = np.zeros(n)
        dx[c] = h
        cp, pp = O.retract(p, cam, p.pts, dx)
        cm, pm = O.retract(p, cam, p.pts, -dx)
        fp, _ = O.residuals(p, cp, pp)
        fm, _ = O.residuals(p, cm, pm)
        fd = (fp - fm) / (2 * h)
        an = J[:, c].toarray().ravel()
        # the robust kernel switches branch at |r| = tau: skip observations within 1% of the threshold
        err = np.abs(fd - an).max() / max(np.abs(an).max(), 1e-12)
        worst = max(worst, err)
    assert worst < 2e-3, worst


def test_jacobian_is_the_derivative_of_the_residual_in_quad(O, ba, small):
    """The independent pin of ora_jacobian (dE_pos, BAFunctor.h:181-297): the derivative of ora_residuals (E_pos, :160-178) through
    update_params (:299-342) by Richardson-extrapolated central differences in __float128 (oracle/ba_referee.c: ref_jacobian_fd),
    on EVERY observation of a synthetic problem and of both BAL files the reference ships.  Measured: 5e-12 of the block's largest
    entry (the double differences above cannot do better than 2e-3: the robust kernel bends on the scale of 0.5 px, a pixel moves
    1e4 px per radian, and steps under the 1e-6 Rodrigues cut-off are not available in double)."""
    from conftest import DATA21, DATA39
    for p in (small, to_oracle(ba.Problem.load_bal(DATA21)), to_oracle(ba.Problem.load_bal(DATA39))):
        cam = O.init_cams(p)
        Jc, Jp = O.jacobian(p, cam, p.pts)
        Jc, Jp = np.asarray(Jc).reshape(p.K, 2, 9), np.asarray(Jp).reshape(p.K, 2, 3)
        Fc, Fp = O.referee_jacobian_fd(p, cam, p.pts)
        sc = np.maximum(np.abs(Fc).max(axis=(1, 2)), np.abs(Fp).max(axis=(1, 2)))
        assert (np.abs(Jc - Fc).max(axis=(1, 2)) / sc).max() < 1e-10
        assert (np.abs(Jp - Fp).max(axis=(1, 2)) / sc).max() < 1e-10



@pytest.mark.parametrize("kind", [2, 1, 0, 3])
def test_step_solves_normal_equations(O, small, kind):
    """All four symbols solve (J'J + lambda I) dx = -J'r (QR of [J; sqrt(lambda) I] == normal equations)."""
    p = small
    cam = O.init_cams(p)
    f, _ = O.residuals(p, cam, p.pts)
    Jc, Jp = O.jacobian(p, cam, p.pts)
    J = dense_J(p, Jc, Jp)
    for lam in (1e-3, 10.0):
        s = O.step(kind, p, Jc, Jp, f, lam)
        g = -(J.T @ f)
        assert np.allclose(s["g"], g, rtol=1e-12, atol=1e-9 * np.abs(g).max())
        H = (J.T @ J + lam * sp.identity(J.shape[1])).tocsc()
        ref = spla.spsolve(H, g)
        assert np.linalg.norm(s["dx"] - ref) < 1e-7 * np.linalg.norm(ref)
        assert np.linalg.norm(H @ s["dx"] - g) < 1e-10 * np.linalg.norm(g)
        # reduced camera matrix == Schur complement of the point block
        M3 = 3 * p.M
        Hd = H.toarray()
        Sref = Hd[M3:, M3:] - Hd[M3:, :M3] @ np.linalg.solve(Hd[:M3, :M3], Hd[:M3, M3:])
        assert np.abs(s["S"] - Sref).max() < 1e-9 * np.abs(Sref).max()
        assert abs(s["diagmax"] - (J.multiply(J)).sum(axis=0).max()) < 1e-12 * s["diagmax"]


def test_whole_matrix_qr_symbol(O, ba):
    """QRSPQR (kind 4): the oracle's Householder QR of the WHOLE [J ; sqrt(lambda) I] (all 3M + 9N columns, natural order, no
    block elimination) against numpy's LAPACK least-squares solve of the same dense matrix and against the block-eliminating
    symbols -- an independent route to the same step."""
    p = to_oracle(ba.Problem.synthetic(5, 60, 200, 19))
    cam = O.init_cams(p)
    f, _ = O.residuals(p, cam, p.pts)
    Jc, Jp = O.jacobian(p, cam, p.pts)
    J = dense_J(p, Jc, Jp).toarray()
    n = J.shape[1]
    for lam in (1e-6, 1e-3, 10.0):
        s = O.step(O.QRSPQR, p, Jc, Jp, f, lam)
        A = np.vstack([J, np.sqrt(lam) * np.eye(n)])
        ref = np.linalg.lstsq(A, -np.concatenate([f, np.zeros(n)]), rcond=None)[0]
        assert np.linalg.norm(s["dx"] - ref) < 1e-7 * np.linalg.norm(ref)  # (cond of the matrix ~1e6 at lambda = 1e-6)
        g = -(J.T @ f)
        assert np.linalg.norm(J.T @ (J @ s["dx"]) + lam * s["dx"] - g) < 1e-10 * np.linalg.norm(g)
        for kind in (O.QRKIT, O.QRCHOL, O.CHOLESKY):
            dx = O.step(kind, p, Jc, Jp, f, lam)["dx"]
            assert np.linalg.norm(dx - s["dx"]) < 1e-7 * np.linalg.norm(s["dx"])
    # the LM loop runs on the symbol (lambda0 = 1e-12 max diag J'J like QRKIT / QRCHOL)
    r4 = O.minimize(O.QRSPQR, p, max_trials=6)["trace"]
    r0 = O.minimize(O.QRKIT, p, max_trials=6)["trace"]
    assert np.array_equal(r4[:, :2], r0[:, :2]) and np.allclose(r4[:3, 2], r0[:3, 2], rtol=1e-7)


def test_three_symbols_agree(O, small):
    p = small
    cam = O.init_cams(p)
    f, _ = O.residuals(p, cam, p.pts)
    Jc, Jp = O.jacobian(p, cam, p.pts)
    ref = O.step(O.CHOLESKY, p, Jc, Jp, f, 1e-4)["dx"]
    for kind in (O.QRCHOL, O.QRKIT, O.MOREQR):
        dx = O.step(kind, p, Jc, Jp, f, 1e-4)["dx"]
        assert np.linalg.norm(dx - ref) < 1e-7 * np.linalg.norm(ref)


def test_rodrigues_cutoff_and_left_multiplication(O, small):
    """Quirks kept: |d omega| <= 1e-6 leaves R untouched (MathUtils.h:74); R <- dR R without re-orthonormalisation."""
    p = small
    cam = O.init_cams(p).reshape(p.N, 15)
    dx = np.zeros(3 * p.M + 9 * p.N)
    dx[3 * p.M + 3: 3 * p.M + 6] = [5e-7, 5e-7, 5e-7]  # norm 8.7e-7 < 1e-6
    dx[3 * p.M + 9 + 3: 3 * p.M + 9 + 6] = [0.1, -0.2, 0.05]
    dx[3 * p.M + 6] = 2.5
    c2, p2 = O.retract(p, cam.reshape(-1), p.pts, dx)
    c2 = c2.reshape(p.N, 15)
    assert np.array_equal(c2[0, :9], cam[0, :9])
    om = np.array([0.1, -0.2, 0.05])
    th = np.linalg.norm(om)
    Kx = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    dR = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
    assert np.allclose(c2[1, :9].reshape(3, 3), dR @ cam[1, :9].reshape(3, 3), rtol=1e-14, atol=1e-15)
    assert c2[0, 12] == cam[0, 12] + 2.5 and np.array_equal(p2, p.pts)


def test_lm_loop_quirks(O, ba):
    """Flat-line exit happens before x = xTest (BacktrackLevMarqQRChol.h:419-428): the returned parameters are those
    of the previous accepted iterate; rejected trials repeat the outer iteration number; lambda >= 1e-10."""
    p = to_oracle(ba.Problem.synthetic(6, 120, 420, 9))
    r = O.minimize(O.CHOLESKY, p, max_trials=400)
    tr = r["trace"]
    assert r["status"] in (0, 1)
    assert tr[:, 4].min() >= 1e-10
    acc = tr[tr[:, 1] == 1]
    assert np.all(np.diff(acc[:, 6]) < 0)  # accepted test energies strictly decrease
    if r["status"] == 0:
        _, e_ret = O.residuals(p, r["cam15"], r["pts"])
        assert abs(e_ret - acc[-1, 2]) < 1e-9 * e_ret  # energy of the iterate BEFORE the last accepted step
    rej = tr[tr[:, 1] == 0]
    if len(rej):
        k = int(np.where(tr[:, 1] == 0)[0][0])
        assert tr[k + 1, 0] == tr[k, 0] and tr[k + 1, 5] == pytest.approx(2 * tr[k, 5])


def test_float_oracle_runs(O, small):
    p = small
    r = O.minimize(O.QRCHOL, p, dtype=np.float32, max_trials=8)
    acc = r["trace"][r["trace"][:, 1] == 1]
    assert len(acc) >= 2 and np.all(np.diff(acc[:, 6]) < 0)
