"""Thin ctypes binding of libba_mi355x.so (include/ba_mi355x.h) for the parity tests and bench.py.

The product is the C-ABI library + the Bundle_Adjustment_{QRKit,QRChol,Cholesky} executables (csrc/main.c); this
module adds nothing to the data path.  It fails loudly when the HIP library is missing -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libba_mi355x.so")

QRKIT, QRCHOL, CHOLESKY, MOREQR, QRSPQR = 0, 1, 2, 3, 4
F64, F32 = 0, 1
KIND_NAMES = {QRKIT: "QRKIT", QRCHOL: "QRCHOL", CHOLESKY: "CHOLESKY", MOREQR: "MOREQR", QRSPQR: "QRSPQR"}
STATUS = {-2: "NotStarted", -1: "Running", 0: "Success", 1: "ExceededLambdaMax", 2: "TooManyFunctionEvaluation",
          3: "MaxItersReached"}

(GET_RESIDUALS, GET_JC, GET_JP, GET_GRAD, GET_S, GET_RHS, GET_DX, GET_CAMS, GET_POINTS, GET_CAMS_TEST,
 GET_POINTS_TEST) = range(11)

EXPORTS = [
    "ba_status_string", "ba_error_string", "ba_problem_load_bal", "ba_problem_create", "ba_problem_synthetic",
    "ba_problem_save_bal", "ba_problem_free", "ba_problem_dims", "ba_problem_get", "ba_lm_params_default",
    "ba_solver_create", "ba_solver_free", "ba_solver_set_allreduce", "ba_solver_set_stream", "ba_solver_shard",
    "ba_minimize", "ba_solver_linearize", "ba_solver_try_step", "ba_solver_accept", "ba_solver_stats", "ba_solver_get",
    "ba_solver_keep_intermediates", "ba_solver_set_state", "ba_solver_timing", "ba_solver_time_phase", "ba_device_info",
    "ba_version", "ba_shard_plan", "ba_problem_save_cache", "ba_problem_load_cache", "ba_solver_selftest",
    "ba_comm_unique_id", "ba_comm_id_via_file", "ba_comm_id_file_done", "ba_solver_comm_init", "ba_solver_recoveries",
]


class LMParams(C.Structure):
    _fields_ = [("lambda_min", C.c_double), ("lambda_max", C.c_double), ("lambda_decrease", C.c_double),
                ("lambda_increase_base", C.c_double), ("lambda_init", C.c_double), ("tol_fun", C.c_double),
                ("max_iter", C.c_int), ("max_fun_ev", C.c_int), ("max_trials", C.c_int), ("verbose", C.c_int)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("trials", C.c_int), ("fun_evals", C.c_int),
                ("energy", C.c_double), ("lambda_", C.c_double), ("seconds", C.c_double), ("schur_ms", C.c_double),
                ("linearize_ms", C.c_double)]


class Timing(C.Structure):
    _fields_ = [("linearize_ms", C.c_double), ("eliminate_ms", C.c_double), ("schur_ms", C.c_double),
                ("factor_ms", C.c_double), ("backsub_ms", C.c_double), ("test_eval_ms", C.c_double),
                ("comm_ms", C.c_double), ("trial_ms", C.c_double), ("n_linearize", C.c_longlong), ("n_trials", C.c_longlong),
                ("n_graph_trials", C.c_longlong)]


TRIAL_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)


class BAError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        RuntimeError.__init__(self, "%s failed: %s (code %d)" % (where, error_string(code), code))


def build(force=False):
    """Compile the HIP library and the executables for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src, "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `make -C %s` (hipcc, gfx950); there is no CPU fallback"
                              % (LIB_PATH, os.path.join(_HERE, "csrc")))
        L = C.CDLL(LIB_PATH)
        L.ba_status_string.restype = C.c_char_p
        L.ba_error_string.restype = C.c_char_p
        L.ba_version.restype = C.c_char_p
        L.ba_problem_free.restype = None
        L.ba_solver_free.restype = None
        L.ba_lm_params_default.restype = None
        L.ba_problem_free.argtypes = [C.c_void_p]
        L.ba_solver_free.argtypes = [C.c_void_p]
        L.ba_problem_synthetic.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_void_p]
        L.ba_solver_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ba_solver_try_step.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ba_solver_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ba_solver_accept.argtypes = [C.c_void_p]
        L.ba_solver_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ba_solver_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.ba_solver_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ba_solver_keep_intermediates.argtypes = [C.c_void_p, C.c_int]
        L.ba_solver_set_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ba_solver_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.ba_solver_shard.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.ba_minimize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ba_solver_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ba_solver_time_phase.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.ba_solver_selftest.argtypes = [C.c_void_p, C.c_int]
        L.ba_comm_unique_id.argtypes = [C.c_void_p]
        L.ba_comm_id_via_file.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.ba_solver_comm_init.argtypes = [C.c_void_p, C.c_void_p]
        L.ba_comm_id_file_done.argtypes = [C.c_char_p, C.c_int]
        L.ba_solver_recoveries.argtypes = [C.c_void_p]
        L.ba_problem_dims.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.ba_problem_get.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.ba_problem_load_bal.argtypes = [C.c_char_p, C.c_void_p]
        L.ba_problem_save_bal.argtypes = [C.c_void_p, C.c_char_p]
        L.ba_problem_create.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.ba_device_info.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ba_shard_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ba_problem_save_cache.argtypes = [C.c_void_p, C.c_char_p]
        L.ba_problem_load_cache.argtypes = [C.c_char_p, C.c_void_p]
        _lib = L
    return _lib


def error_string(code):
    return lib().ba_error_string(code).decode()


def status_string(status):
    return lib().ba_status_string(status).decode()


def _chk(rc, where):
    if rc != 0:
        raise BAError(rc, where)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Problem:
    """ba_problem handle: the BAL problem as the reference's loader reads it (bundle_adjustment_large.cpp:59-107)."""

    def __init__(self, handle):
        self._h = handle
        N, M, K = C.c_int(), C.c_int(), C.c_int()
        _chk(lib().ba_problem_dims(self._h, C.byref(N), C.byref(M), C.byref(K)), "ba_problem_dims")
        self.N, self.M, self.K = N.value, M.value, K.value

    @classmethod
    def load_bal(cls, path):
        h = C.c_void_p()
        _chk(lib().ba_problem_load_bal(str(path).encode(), C.byref(h)), "ba_problem_load_bal(%s)" % path)
        return cls(h)

    @classmethod
    def synthetic(cls, N, M, K, seed):
        h = C.c_void_p()
        _chk(lib().ba_problem_synthetic(N, M, K, seed, C.byref(h)), "ba_problem_synthetic")
        return cls(h)

    @classmethod
    def from_arrays(cls, N, M, K, cam_idx, pt_idx, meas, cams9, pts):
        cam_idx = np.ascontiguousarray(cam_idx, np.int32)
        pt_idx = np.ascontiguousarray(pt_idx, np.int32)
        meas = np.ascontiguousarray(meas, np.float64)
        cams9 = np.ascontiguousarray(cams9, np.float64)
        pts = np.ascontiguousarray(pts, np.float64)
        h = C.c_void_p()
        _chk(lib().ba_problem_create(N, M, K, _p(cam_idx), _p(pt_idx), _p(meas), _p(cams9), _p(pts), C.byref(h)),
             "ba_problem_create")
        return cls(h)

    def arrays(self):
        cam_idx = np.empty(self.K, np.int32)
        pt_idx = np.empty(self.K, np.int32)
        meas = np.empty(2 * self.K)
        cams9 = np.empty(9 * self.N)
        pts = np.empty(3 * self.M)
        _chk(lib().ba_problem_get(self._h, _p(cam_idx), _p(pt_idx), _p(meas), _p(cams9), _p(pts)), "ba_problem_get")
        return dict(cam_idx=cam_idx, pt_idx=pt_idx, meas=meas, cams9=cams9, pts=pts)

    def shard_plan(self, rank, world):
        out = (C.c_longlong * 8)()
        _chk(lib().ba_shard_plan(self._h, rank, world, out), "ba_shard_plan")
        keys = ("p0", "p1", "o0", "o1", "entries", "chunks", "pairs", "was_sorted")
        return dict(zip(keys, [int(v) for v in out]))

    @classmethod
    def load_cache(cls, path):
        h = C.c_void_p()
        _chk(lib().ba_problem_load_cache(str(path).encode(), C.byref(h)), "ba_problem_load_cache(%s)" % path)
        return cls(h)

    def save_cache(self, path):
        _chk(lib().ba_problem_save_cache(self._h, str(path).encode()), "ba_problem_save_cache")

    def save_bal(self, path):
        _chk(lib().ba_problem_save_bal(self._h, str(path).encode()), "ba_problem_save_bal")

    @property
    def D(self):
        return 9 * self.N

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ba_problem_free(self._h)
            self._h = None


class Solver:
    """ba_solver handle: device-resident LM state of one shard."""

    def __init__(self, problem, kind=CHOLESKY, scalar=F64, device=-1, shard_rank=0, shard_world=1):
        self.problem = problem
        self.kind, self.scalar = kind, scalar
        h = C.c_void_p()
        _chk(lib().ba_solver_create(problem._h, kind, scalar, device, shard_rank, shard_world, C.byref(h)),
             "ba_solver_create")
        self._h = h
        v = [C.c_int() for _ in range(4)]
        _chk(lib().ba_solver_shard(self._h, *[C.byref(x) for x in v]), "ba_solver_shard")
        self.p0, self.p1, self.o0, self.o1 = [x.value for x in v]
        self.Ml, self.Kl = self.p1 - self.p0, self.o1 - self.o0
        self._cb_keep = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ba_solver_free(self._h)
            self._h = None

    # -- seam -------------------------------------------------------------------------------------------
    def linearize(self, want_diag_max=True):
        e, d = C.c_double(), C.c_double()
        _chk(lib().ba_solver_linearize(self._h, C.byref(e), C.byref(d) if want_diag_max else None), "ba_solver_linearize")
        return e.value, d.value

    def try_step(self, lam):
        e, r, n = C.c_double(), C.c_double(), C.c_double()
        _chk(lib().ba_solver_try_step(self._h, float(lam), C.byref(e), C.byref(r), C.byref(n)), "ba_solver_try_step")
        return e.value, r.value, n.value

    def accept(self):
        _chk(lib().ba_solver_accept(self._h), "ba_solver_accept")

    def stats(self):
        out = np.empty(4)
        _chk(lib().ba_solver_stats(self._h, _p(out)), "ba_solver_stats")
        return dict(mean_err=out[0], inlier_mean_err=out[1], n_inliers=int(out[2]), objective=out[3])

    def keep_intermediates(self, on=True):
        _chk(lib().ba_solver_keep_intermediates(self._h, int(on)), "ba_solver_keep_intermediates")

    def get(self, what):
        N, D, Ml, Kl = self.problem.N, self.problem.D, self.Ml, self.Kl
        n = {GET_RESIDUALS: 2 * Kl, GET_JC: 18 * Kl, GET_JP: 6 * Kl, GET_GRAD: 3 * Ml + D, GET_S: D * D, GET_RHS: D,
             GET_DX: 3 * Ml + D, GET_CAMS: 15 * N, GET_POINTS: 3 * Ml, GET_CAMS_TEST: 15 * N, GET_POINTS_TEST: 3 * Ml}[what]
        out = np.empty(n)
        _chk(lib().ba_solver_get(self._h, what, _p(out), n), "ba_solver_get(%d)" % what)
        if what == GET_S:
            return out.reshape(D, D).T
        return out

    def set_state(self, cam15=None, pts=None):
        cam15 = None if cam15 is None else np.ascontiguousarray(cam15, np.float64)
        pts = None if pts is None else np.ascontiguousarray(pts, np.float64)
        _chk(lib().ba_solver_set_state(self._h, _p(cam15), _p(pts)), "ba_solver_set_state")

    def minimize(self, max_trials=0, verbose=False, trace=True, **lm_over):
        lm = LMParams()
        lib().ba_lm_params_default(C.byref(lm))
        lm.max_trials, lm.verbose = int(max_trials), int(verbose)
        for k, v in lm_over.items():
            setattr(lm, k, v)
        rows = []
        cb = TRIAL_CB(lambda user, it, acc, f, rho, lam, el: rows.append((it, acc, f, rho, lam, el))) if trace else None
        res = Result()
        _chk(lib().ba_minimize(self._h, C.byref(lm), cb, None, C.byref(res)), "ba_minimize")
        return dict(status=res.status, iterations=res.iterations, trials=res.trials, fun_evals=res.fun_evals,
                    energy=res.energy, lam=res.lambda_, seconds=res.seconds, schur_ms=res.schur_ms,
                    linearize_ms=res.linearize_ms, trace=np.array(rows).reshape(-1, 6))

    def timing(self, reset=False):
        t = Timing()
        _chk(lib().ba_solver_timing(self._h, C.byref(t), int(reset)), "ba_solver_timing")
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def time_phase(self, phase, reps, lam):
        ms = C.c_double()
        _chk(lib().ba_solver_time_phase(self._h, phase, reps, float(lam), C.byref(ms)), "ba_solver_time_phase")
        return ms.value

    def recoveries(self):
        """Trials ba_minimize repeated through the launch-per-step factorisation after a hand-off time-out."""
        return int(lib().ba_solver_recoveries(self._h))

    def selftest(self, which):
        """Returns the library's return code (not raised): the failure paths are what this hook exists to show."""
        return int(lib().ba_solver_selftest(self._h, int(which)))

    # -- multi-GPU plumbing -------------------------------------------------------------------------------
    def set_stream(self, raw_stream):
        _chk(lib().ba_solver_set_stream(self._h, C.c_void_p(raw_stream)), "ba_solver_set_stream")

    def comm_init(self, comm_id):
        """RCCL inside the library: comm_id = the 128 bytes of comm_unique_id() from shard rank 0; collective over the shard group."""
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        _chk(lib().ba_solver_comm_init(self._h, buf), "ba_solver_comm_init")

    def set_allreduce(self, pyfunc):
        """pyfunc(dev_ptr:int, count:int, scalar:int, op:int, stream:int) -> int (0 = ok)."""
        def tramp(user, buf, count, scalar, op, stream):
            try:
                return int(pyfunc(buf or 0, count, scalar, op, stream or 0))
            except Exception as exc:  # never unwind through C
                import sys
                print("allreduce callback raised: %r" % (exc,), file=sys.stderr)
                return 1
        self._cb_keep = ALLREDUCE_FN(tramp)
        _chk(lib().ba_solver_set_allreduce(self._h, C.cast(self._cb_keep, C.c_void_p), None), "ba_solver_set_allreduce")


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the library (call on shard rank 0, carry the bytes to the other ranks)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(lib().ba_comm_unique_id(buf), "ba_comm_unique_id")
    return buf.raw


def device_info(device=-1):
    name = C.create_string_buffer(256)
    cus = C.c_int()
    _chk(lib().ba_device_info(device, name, 256, C.byref(cus)), "ba_device_info")
    return name.value.decode(), cus.value
