"""ctypes binding of the CPU oracle (oracle/libba_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The oracle restates the reference algorithm (see oracle/ba_oracle_impl.h for file:line citations).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libba_oracle.so")

QRKIT, QRCHOL, CHOLESKY, MOREQR, QRSPQR = 0, 1, 2, 3, 4  # (QRSPQR: dense Householder QR of the WHOLE [J; sqrt(lambda) I] -- small problems)
LM_DEFAULTS = (1e-10, 1e10, 2.0, 1e-8)  # lambda min/max, increaseBase, tolFun (BacktrackLevMarqQRChol.h:131-146)


def build():
    src = [os.path.join(_ROOT, "oracle", f) for f in ("ba_oracle.c", "ba_oracle_impl.h", "ba_referee.c", "Makefile")]
    sos = [_SO, os.path.join(_ROOT, "oracle", "libba_referee.so")]
    if all(os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in src) for so in sos):
        return _SO
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ora_residuals_f64.restype = C.c_double
        _lib.ora_residuals_f32.restype = C.c_float
        _lib.ora_set_threads(1)  # the reference is single-threaded; bench.py's "all cores" column raises it explicitly
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_wide_sums(on):
    """Experiment switch of oracle/ba_oracle_impl.h (build_reduced): S, rhs and g_c accumulated in long double.  Default off."""
    lib().ora_set_wide_sums(int(bool(on)))


def set_more_qr(on):
    """MOREQR's right block QR only (ba_oracle_impl.h: solve_more_qr) (the default since round 4) or, on=False, the LDL^T of S -- the counterpart of BA_MOREQR_QR=0."""
    lib().ora_set_more_qr(int(bool(on)))


def set_threads(n):
    """OpenMP threads of the dense factorisation (results do not depend on it)."""
    lib().ora_set_threads(int(n))


class Problem:
    """BAL problem as flat arrays (bundle_adjustment_large.cpp:59-107)."""

    def __init__(self, N, M, K, cam_idx, pt_idx, meas, cams9, pts):
        self.N, self.M, self.K = int(N), int(M), int(K)
        self.cam_idx = np.ascontiguousarray(cam_idx, np.int32)
        self.pt_idx = np.ascontiguousarray(pt_idx, np.int32)
        self.meas = np.ascontiguousarray(meas, np.float64).reshape(-1)
        self.cams9 = np.ascontiguousarray(cams9, np.float64).reshape(-1)
        self.pts = np.ascontiguousarray(pts, np.float64).reshape(-1)

    @property
    def D(self):
        return 9 * self.N

    def subset(self, n_points):
        """First n_points points and their observations (cameras unchanged) -- a small excerpt."""
        k = int(np.searchsorted(self.pt_idx, n_points, side="left"))
        return Problem(self.N, n_points, k, self.cam_idx[:k], self.pt_idx[:k], self.meas[: 2 * k], self.cams9,
                       self.pts[: 3 * n_points])


def ensemble_member(p, k, rel=1e-13):
    """Member k of the 1e-13 ensembles (tests/golden/make_referee.py, tests/test_gpu_configs.py): the problem with cameras and points
    perturbed by rel x N(0,1) relative, seeded 1000 + k; member 0 is the unperturbed input.  The SAME inputs go to the oracle and the GPU."""
    if k == 0:
        return p
    rng = np.random.default_rng(1000 + k)
    cams9 = p.cams9 * (1 + rel * rng.standard_normal(p.cams9.shape))
    pts = p.pts * (1 + rel * rng.standard_normal(p.pts.shape))
    return Problem(p.N, p.M, p.K, p.cam_idx, p.pt_idx, p.meas, cams9, pts)


def load_bal(path):
    L = lib()
    N, M, K = C.c_int(), C.c_int(), C.c_int()
    rc = L.ora_bal_header(path.encode(), C.byref(N), C.byref(M), C.byref(K))
    if rc:
        raise IOError("ora_bal_header rc=%d for %s" % (rc, path))
    N, M, K = N.value, M.value, K.value
    cam_idx = np.empty(K, np.int32)
    pt_idx = np.empty(K, np.int32)
    meas = np.empty(2 * K)
    cams9 = np.empty(9 * N)
    pts = np.empty(3 * M)
    rc = L.ora_bal_read(path.encode(), N, M, K, _p(cam_idx), _p(pt_idx), _p(meas), _p(cams9), _p(pts))
    if rc:
        raise IOError("ora_bal_read rc=%d for %s" % (rc, path))
    return Problem(N, M, K, cam_idx, pt_idx, meas, cams9, pts)


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64", C.c_double
    if dtype == np.float32:
        return "_f32", C.c_float
    raise ValueError(dtype)


def init_cams(p, dtype=np.float64):
    sfx, _ = _sfx(dtype)
    cam15 = np.empty(15 * p.N, dtype)
    getattr(lib(), "ora_init_cams" + sfx)(p.N, _p(p.cams9), _p(cam15))
    return cam15


def residuals(p, cam15, pts, tau=0.5):
    sfx, ct = _sfx(cam15.dtype)
    f = np.empty(2 * p.K, cam15.dtype)
    meas = p.meas.astype(cam15.dtype)
    e = getattr(lib(), "ora_residuals" + sfx)(p.N, p.M, p.K, _p(cam15), _p(pts), _p(p.cam_idx), _p(p.pt_idx), _p(meas),
                                              ct(tau), _p(f))
    return f, float(e)


def jacobian(p, cam15, pts, tau=0.5):
    sfx, ct = _sfx(cam15.dtype)
    Jc = np.empty(18 * p.K, cam15.dtype)
    Jp = np.empty(6 * p.K, cam15.dtype)
    meas = p.meas.astype(cam15.dtype)
    getattr(lib(), "ora_jacobian" + sfx)(p.N, p.M, p.K, _p(cam15), _p(pts), _p(p.cam_idx), _p(p.pt_idx), _p(meas), ct(tau),
                                         _p(Jc), _p(Jp))
    return Jc.reshape(p.K, 2, 9), Jp.reshape(p.K, 2, 3)


def retract(p, cam15, pts, dx):
    sfx, _ = _sfx(cam15.dtype)
    co = np.empty_like(cam15)
    po = np.empty_like(pts)
    dx = np.ascontiguousarray(dx, cam15.dtype)
    getattr(lib(), "ora_retract" + sfx)(p.N, p.M, _p(cam15), _p(pts), _p(dx), _p(co), _p(po))
    return co, po


def stats(p, cam15, pts, tau=0.5):
    sfx, ct = _sfx(cam15.dtype)
    out = np.empty(4)
    meas = p.meas.astype(cam15.dtype)
    getattr(lib(), "ora_stats" + sfx)(p.N, p.M, p.K, _p(cam15), _p(pts), _p(p.cam_idx), _p(p.pt_idx), _p(meas), ct(tau),
                                      _p(out))
    return dict(mean_err=out[0], inlier_mean_err=out[1], n_inliers=int(out[2]), objective=out[3])


ASSEMBLE_ONLY = 256  # or-ed into kind: elimination + reduced system only (no dense factorisation, dx = 0)


def step(kind, p, Jc, Jp, fvec, lam, want_S=True):
    dt = fvec.dtype
    sfx, ct = _sfx(dt)
    D = p.D
    dx = np.zeros(3 * p.M + D, dt)
    S = np.empty(D * D, dt) if want_S else None
    rhs = np.empty(D, dt)
    g = np.empty(3 * p.M + D, dt)
    dmax = ct(0)
    Jc = np.ascontiguousarray(Jc, dt).reshape(-1)
    Jp = np.ascontiguousarray(Jp, dt).reshape(-1)
    rc = getattr(lib(), "ora_step" + sfx)(kind, p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(Jc), _p(Jp), _p(fvec),
                                          ct(lam), _p(dx), _p(S), _p(rhs), _p(g), C.byref(dmax))
    if rc:
        raise RuntimeError("ora_step rc=%d" % rc)
    return dict(dx=dx, S=None if S is None else S.reshape(D, D).T, rhs=rhs, g=g, diagmax=float(dmax.value))


TRACE_COLS = ("iter", "accepted", "f", "rho", "lambda", "lambda_used", "e_test", "dx_norm")


def minimize(kind, p, dtype=np.float64, max_trials=1000000, lm=LM_DEFAULTS, max_iter=1000000, max_fun_ev=1000000, tau=0.5,
             cam15=None, pts=None, snapshots=False):
    """snapshots=True: also returns 'snap' (trials x (15N + 3M) doubles), the state x every trial started from."""
    sfx, ct = _sfx(dtype)
    cam15 = init_cams(p, dtype) if cam15 is None else cam15.copy()
    pts = p.pts.astype(dtype) if pts is None else pts.copy()
    meas = p.meas.astype(dtype)
    cap = min(max_trials, 100000)
    trace = np.zeros((cap, 8))
    lmv = np.asarray(lm, np.float64)
    ntr = C.c_int(0)
    snap = np.zeros((cap, 15 * p.N + 3 * p.M)) if snapshots else None
    status = getattr(lib(), "ora_minimize" + sfx)(kind, p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(meas), ct(tau),
                                                  _p(cam15), _p(pts), _p(lmv), max_iter, max_fun_ev, cap, _p(trace),
                                                  C.byref(ntr), _p(snap))
    out = dict(status=status, trace=trace[: ntr.value], cam15=cam15, pts=pts)
    if snapshots:
        out["snap"] = snap[: ntr.value]
    return out


_ref = None


def referee():
    """oracle/libba_referee.so: the same restatement in __float128 (ba_referee.c)."""
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(os.path.join(_ROOT, "oracle", "libba_referee.so"))
        _ref.ref_energy.restype = C.c_double
    return _ref


REF_COLS = ("energy", "e_test", "rho_scale", "dx_norm", "diagmax", "grad_norm", "backward_error", "spare")


def referee_trial(kind, p, cam15, pts, lam, tau=0.5, want_dx=False):
    """One LM trial in quad precision from the double state (cam15, pts, lam); see oracle/ba_referee.c."""
    cam15 = np.ascontiguousarray(cam15, np.float64).reshape(-1)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1)
    out = np.zeros(8)
    dx = np.zeros(3 * p.M + 9 * p.N) if want_dx else None
    rc = referee().ref_trial(kind, p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(p.meas), C.c_double(tau), _p(cam15), _p(pts),
                             C.c_double(lam), _p(out), _p(dx))
    if rc:
        raise RuntimeError("ref_trial rc=%d" % rc)
    r = dict(zip(REF_COLS, out))
    if want_dx:
        r["dx"] = dx
    return r


def referee_minimize(kind, p, max_trials=100000, lm=LM_DEFAULTS, max_iter=1000000, max_fun_ev=1000000, tau=0.5, x87=False):
    """The LM loop in quad precision from the problem's double start (oracle/ba_referee.c: ref_minimize) -- minutes for problem-21;
    x87: in long double instead (ref80_minimize: eps 1.1e-19, ~3x an fp64 run)."""
    cam15 = init_cams(p)
    pts = p.pts.copy()
    trace = np.zeros((max_trials, 8))
    lmv = np.asarray(lm, np.float64)
    ntr = C.c_int(0)
    status = (referee().ref80_minimize if x87 else referee().ref_minimize)(kind, p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(p.meas), C.c_double(tau), _p(cam15), _p(pts),
                                    _p(lmv), max_iter, max_fun_ev, max_trials, _p(trace), C.byref(ntr))
    return dict(status=status, trace=trace[: ntr.value], cam15=cam15, pts=pts)


def referee_reduced(kind, p, cam15, pts, lam, tau=0.5):
    """Reduced camera matrix (D x D, symmetric) and rhs of one trial assembled in quad precision, rounded to double."""
    cam15 = np.ascontiguousarray(cam15, np.float64).reshape(-1)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1)
    D = p.D
    S = np.empty(D * D)
    rhs = np.empty(D)
    rc = referee().ref_reduced(kind, p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(p.meas), C.c_double(tau), _p(cam15), _p(pts),
                               C.c_double(lam), _p(S), _p(rhs))
    if rc:
        raise RuntimeError("ref_reduced rc=%d" % rc)
    return S.reshape(D, D).T, rhs


def referee_energy(p, cam15, pts, tau=0.5):
    cam15 = np.ascontiguousarray(cam15, np.float64).reshape(-1)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1)
    return float(referee().ref_energy(p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(p.meas), C.c_double(tau), _p(cam15), _p(pts)))


def referee_jacobian_fd(p, cam15, pts, tau=0.5):
    """(Jc [K,2,9], Jp [K,2,3]): the derivative of the RESIDUAL function through the retraction, by Richardson-extrapolated central
    differences in quad precision (oracle/ba_referee.c: ref_jacobian_fd) -- independent of the hand-written Jacobian."""
    cam15 = np.ascontiguousarray(cam15, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    Jc = np.zeros((p.K, 2, 9))
    Jp = np.zeros((p.K, 2, 3))
    rc = referee().ref_jacobian_fd(p.N, p.M, p.K, _p(p.cam_idx), _p(p.pt_idx), _p(p.meas), C.c_double(tau), _p(cam15), _p(pts),
                                   _p(Jc), _p(Jp))
    if rc:
        raise RuntimeError("ref_jacobian_fd failed: %d" % rc)
    return Jc, Jp
