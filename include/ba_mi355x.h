/*
 * ba_mi355x.h -- C ABI of libba_mi355x.so: the MI355X (gfx950) Levenberg-Marquardt bundle-adjustment hot path
 * behind the executable / solver-symbol interface of jasvob/BundleAdjustment_Benchmarks.
 *
 * The reference has no FFI: its one seam is `lm.minimize(params)` called from main()
 * (src/bundle_adjustment_large.cpp:130-165) plus the functor calls the LM classes make
 * (src/Eigen_ext/BacktrackLevMarqQRChol.h:213,216-217,257,264,365,368).  Each entry point below names the
 * reference interface it replaces.  Plain pointers and sizes only; no torch / HIP types in signatures
 * (streams travel as void*).
 *
 * All functions return 0 (BA_OK) on success or a BA_ERR_* code; nothing throws across the boundary.
 * A handle is not thread-safe; one host thread drives one solver (the reference is single-threaded).
 */
#ifndef BA_MI355X_H
#define BA_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- codes ------------------------------------------------------------------------------------------- */

/* ReturnCodes of the reference executable (bundle_adjustment_large.cpp:26-31) + library errors. */
enum {
    BA_OK = 0,
    BA_ERR_USAGE = 1, /* WrongInputParams */
    BA_ERR_FILE = 2,  /* WrongInputFile   */
    BA_ERR_PARSE = 3, /* new: the reference does not check stream errors */
    BA_ERR_ARG = 4,
    BA_ERR_HIP = 5,   /* HIP runtime error or no gfx950 device: the product path never falls back to a CPU */
    BA_ERR_NOMEM = 6,
    BA_ERR_COMM = 7
};

/* Solver symbols of the reference build (src/CMakeLists.txt:95-178; src/Optimization/BAFunctor.h:98-117).
 * BA_MOREQR (src/Eigen_ext/BacktrackLevMarqMore.h): two QR factorisations per step -- J once per outer iteration,
 * [R ; sqrt(lambda) I] per trial -- and lambda0 = 1e-6 * max column norm of J.  Both are QR all the way (:288-345): per-point
 * Householder QRs for the point columns, dense Householder QRs for the camera columns (J2bot(lambda = 0) per outer iteration,
 * [rows left by the per-point 6 x 3 QRs ; R22 ; sqrt(lambda) I] per trial) -- no normal equations; needs (6 M + 2 D)(D + 1) scalars
 * of device memory (BA_ERR_NOMEM beyond).  Environment BA_MOREQR_QR=0 at solver creation: the right block by LDL^T of the reduced
 * camera system instead (rounds 1 - 3's variant). */
/* BA_QRSPQR (SuiteSparseQR on the whole [J ; sqrt(lambda) I], BAFunctor.h:113-116, bundle_adjustment_large.cpp:151-157, README.md:17;
 * the library itself is absent): a sparse QR of this matrix under a fill-reducing column ordering eliminates the 3-column point
 * blocks first and is left with one dense front, J2bot -- which is the factorisation the QRKIT path performs, so the symbol runs
 * that path (per-point Householder QR, dense Householder QR of J2bot); same LM loop as QRKIT (Eigen::BacktrackLevMarq).  The
 * equivalence is tested against a whole-matrix Householder QR with no block elimination (tests: test_qrspqr_against_the_whole_matrix_qr).
 * Sharded (shard_world > 1), BA_QRKIT and BA_QRSPQR keep their dense QR: every shard factors its own rows of J2bot, the exchange step
 * sums a zeroed stack into which each shard has put its D x D triangle R (+ the head of Q^T rhs, g_c, energy), and the QR of the stack
 * runs redundantly (distributed TSQR) -- never the normal equations these symbols exist to avoid. */
typedef enum { BA_QRKIT = 0, BA_QRCHOL = 1, BA_CHOLESKY = 2, BA_MOREQR = 3, BA_QRSPQR = 4 } ba_solver_kind;

/* `typedef double Scalar;` / `typedef float Scalar;` (src/BATypeUtils.h:6-7). */
typedef enum { BA_F64 = 0, BA_F32 = 1 } ba_scalar;

/* BacktrackLevMarq*Info::Status (BacktrackLevMarqQRChol.h:39-46, BacktrackLevMarqCholesky.h:27-34). */
typedef enum {
    BA_NOT_STARTED = -2,
    BA_RUNNING = -1, /* also returned when the max_trials extension stopped the loop */
    BA_SUCCESS = 0,
    BA_EXCEEDED_LAMBDA_MAX = 1,
    BA_TOO_MANY_FUN_EVALS = 2,
    BA_MAX_ITERS = 3
} ba_status;

/* statusToString (BacktrackLevMarqQRChol.h:48-63). */
const char *ba_status_string(int status);
const char *ba_error_string(int err);

/* ---- problem: BAL loader (bundle_adjustment_large.cpp:59-107) ------------------------------------------ */

typedef struct ba_problem ba_problem;

/* Parses `N M K`, K x `cam pt u v`, 9N camera scalars (omega(3), T(3), f, k1, k2), 3M point scalars.
 * BA_ERR_FILE if the file cannot be opened (reference: "Cannot open <path>", exit 2). */
int ba_problem_load_bal(const char *path, ba_problem **out);
/* Same problem from caller-owned arrays (copied). meas is 2K interleaved (u,v); cams9 is 9N; pts is 3M. */
int ba_problem_create(int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas,
                      const double *cams9, const double *pts, ba_problem **out);
/* Seeded synthetic BAL problem with exactly (N, M, K) (stand-in for the data files missing from the reference
 * checkout and for the 1024-camera scaling config; generator described in DESIGN.md). mean_obs is informative only. */
int ba_problem_synthetic(int N, int M, int K, unsigned long long seed, ba_problem **out);
int ba_problem_save_bal(const ba_problem *p, const char *path);
/* Binary cache of a parsed problem (`ifstream >>` / strtod parsing of a 280 MB text file takes seconds; the
 * cache is the same arrays, little-endian, behind a 32-byte header) -- an extension, the BAL text format stays the interface. */
int ba_problem_save_cache(const ba_problem *p, const char *path);
int ba_problem_load_cache(const char *path, ba_problem **out);
void ba_problem_free(ba_problem *p);
int ba_problem_dims(const ba_problem *p, int *N, int *M, int *K);
/* Copies out the arrays (any pointer may be NULL). */
int ba_problem_get(const ba_problem *p, int *cam_idx, int *pt_idx, double *meas, double *cams9, double *pts);

/* Host-only view of the static structure ba_solver_create would build for shard `rank` of `world` (no GPU needed):
 * out8 = {p0, p1, o0, o1, entries, chunks, camera pairs, input_was_sorted_by_point}.  Points are split into contiguous
 * ranges balanced by observation count; entries are the (point, camera pair) contributions to the reduced matrix. */
int ba_shard_plan(const ba_problem *p, int shard_rank, int shard_world, long long *out8);

/* ---- solver: device-resident LM state ------------------------------------------------------------------ */

typedef struct ba_solver ba_solver;

/* LMParams + Lambda (BacktrackLevMarqQRChol.h:124-146).  max_trials / verbose are extensions (0 = reference). */
typedef struct {
    double lambda_min;           /* 1e-10 */
    double lambda_max;           /* 1e10  */
    double lambda_decrease;      /* 10 (unused by the reference loop) */
    double lambda_increase_base; /* 2     */
    double lambda_init;          /* 1e-3 (overwritten by 1e-12*max diag(J'J) at iteration 1) */
    double tol_fun;              /* 1e-8  */
    int max_iter;                /* 1e6   */
    int max_fun_ev;              /* 1e6   */
    int max_trials;              /* extension: stop after this many table rows (0 = unlimited) */
    int verbose;                 /* print the reference's iteration table to stdout */
} ba_lm_params;
void ba_lm_params_default(ba_lm_params *p);

typedef struct {
    int status;          /* ba_status */
    int iterations;      /* outer iterations started */
    int trials;          /* table rows (accepted + rejected) */
    int fun_evals;
    double energy;       /* m_energy at exit */
    double lambda;
    double seconds;      /* wall time of minimize */
    double schur_ms;     /* mean device time per trial of elimination + Schur assembly + reduced solve + back-substitution */
    double linearize_ms; /* mean device time per outer iteration of residual + Jacobian + gradient */
} ba_result;

/* One row of the reference's table (outputIter, BacktrackLevMarqQRChol.h:84-93): f is the energy BEFORE the step. */
typedef void (*ba_trial_cb)(void *user, int iter, int accepted, double f, double rho, double lambda, double elapsed_s);

/* A collective on `count` scalars of type `scalar` in DEVICE memory across the ranks that shard one problem, in place; called on
 * the host thread between kernels, `stream` is the hipStream_t the solver enqueues on (the callback must order itself against
 * that stream).  op (low byte; round 4 added the last two for the distributed factor, BA_DIST_FACTOR):
 *   BA_OP_SUM / BA_OP_MAX    all-reduce;
 *   BA_OP_BCAST | root << 8  broadcast of dev_buf[0 .. count) from rank `root`;
 *   BA_OP_REDUCE_SCATTER     dev_buf holds shard_world chunks of `count` scalars; on return chunk `shard_rank` of THIS rank's buffer
 *                            is the sum over the ranks of their chunk `shard_rank` (ncclReduceScatter's in-place form; the other
 *                            chunks are unspecified).
 * A transport supplied by the host layer -- used by the gloo tests; the production transport is RCCL inside the library
 * (ba_solver_comm_init: ncclAllReduce / ncclBroadcast / ncclReduceScatter).  Never called when shard_world == 1. */
#define BA_OP_SUM 0
#define BA_OP_MAX 1
#define BA_OP_BCAST 2
#define BA_OP_REDUCE_SCATTER 3
typedef int (*ba_allreduce_fn)(void *user, void *dev_buf, size_t count, int scalar, int op, void *stream);

/* ---- communication of a sharded solve (no reference counterpart: the reference is one process) ------------------------- */

/* RCCL inside the library.  ba_comm_unique_id: rank 0 creates the 128-byte id of a new communicator (ncclGetUniqueId); the host
 * layer carries it to the other ranks (torch.distributed / MPI / a file: ba_comm_id_via_file); then EVERY rank of the shard
 * group calls ba_solver_comm_init (collective: ncclCommInitRank with the solver's shard_rank / shard_world, on the solver's
 * device).  From then on the per-trial all-reduces of the packed reduced camera system and of the step scalars are
 * ncclAllReduce calls on the solver's stream, enqueued by ba_minimize without a host synchronisation. */
#define BA_COMM_ID_BYTES 128
int ba_comm_unique_id(void *id_out /* BA_COMM_ID_BYTES */);
/* Rendezvous through a file for processes started by hand: rank 0 removes whatever an earlier run left at `path`, creates the id and
 * publishes it (O_EXCL temporary file, mode 0600, atomic rename); the others wait up to BA_COMM_WAIT_S (60) seconds for a file that
 * belongs to THIS launch: the nonce of the environment variable BA_COMM_NONCE when the launcher sets one; else a file written after
 * the reader's PROCESS start (library load, less two seconds) at once, an older one only after it has stayed unchanged for
 * BA_COMM_GRACE_S (5) seconds -- this launch's rank 0 would have removed a dead run's leftover at its own start.  ba_comm_id_file_done: call after ba_solver_comm_init has returned (it is collective: every rank has the id
 * by then); rank 0 removes the file, so that no id outlives its launch. */
int ba_comm_id_via_file(const char *path, int rank, void *id_out);
int ba_comm_id_file_done(const char *path, int rank);

/* Replaces the construction of BAFunctor + the LM object (bundle_adjustment_large.cpp:117-131): copies the problem
 * to HBM in SoA layout, builds the static camera-pair structure.  Points (and their observations) are partitioned
 * into shard_world contiguous ranges balanced by observation count; this handle owns range shard_rank.
 * device < 0 keeps the current HIP device. Fails with BA_ERR_HIP when no GPU is present, with BA_ERR_ARG when a QR symbol
 * meets a point with more than 1024 observations (the per-point QR keeps a track in registers; CHOLESKY has no limit). */
int ba_solver_create(const ba_problem *p, ba_solver_kind kind, ba_scalar scalar, int device, int shard_rank,
                     int shard_world, ba_solver **out);
void ba_solver_free(ba_solver *s);
int ba_solver_set_allreduce(ba_solver *s, ba_allreduce_fn fn, void *user); /* host-language transport (gloo tests); RCCL: next line */
int ba_solver_comm_init(ba_solver *s, const void *id /* BA_COMM_ID_BYTES, from ba_comm_unique_id on rank 0 */);
/* Enqueue on a caller-owned hipStream_t (e.g. torch's current stream) instead of the solver's own stream. */
int ba_solver_set_stream(ba_solver *s, void *hip_stream);
/* Point range [p0,p1) and observation range [o0,o1) owned by this shard. */
int ba_solver_shard(const ba_solver *s, int *p0, int *p1, int *o0, int *o1);

/* lm.minimize(params) (bundle_adjustment_large.cpp:134,141,162): runs the LM loop of the solver symbol on the GPU,
 * parameters stay resident and are updated in place (quirk kept: the flat-line exit happens before x = xTest). */
int ba_minimize(ba_solver *s, const ba_lm_params *lm, ba_trial_cb cb, void *user, ba_result *out);

/* Step-level seam = what the LM classes call on the functor / solver, for parity tests and external LM drivers:
 *   linearize : m_functor(x, r); energy; m_functor.df(x, J); JtRes; column norms (BacktrackLevMarqQRChol.h:257-280)
 *   try_step  : m_solver.compute ... dx; xTest = x (+) dx; m_functor(xTest); rhoScale (BacktrackLevMarqQRChol.h:291-375)
 *   accept    : x = xTest (BacktrackLevMarqQRChol.h:428) */
int ba_solver_linearize(ba_solver *s, double *energy, double *diag_max);
int ba_solver_try_step(ba_solver *s, double lambda, double *energy_test, double *rho_scale, double *dx_norm);
int ba_solver_accept(ba_solver *s);
/* Utils::showErrorStatistics + showObjective (Utils.h:15-68) on the resident parameters:
 * out4 = {mean reprojection error, inlier mean error, nInliers, "True objective"}. */
int ba_solver_stats(ba_solver *s, double *out4);

/* Copy device arrays to the host as doubles (own shard only for per-observation / per-point arrays). */
typedef enum {
    BA_GET_RESIDUALS = 0, /* 2K  obs-major interleaved, order of the input file */
    BA_GET_JC = 1,        /* 18K per obs 2x9 row-major, columns [T, omega, f, k1, k2] */
    BA_GET_JP = 2,        /* 6K  per obs 2x3 row-major */
    BA_GET_GRAD = 3,      /* 3M+9N  = -J'r, points first */
    BA_GET_S = 4,         /* D*D column-major reduced camera matrix (full symmetric) of the last try_step */
    BA_GET_RHS = 5,       /* D reduced right-hand side */
    BA_GET_DX = 6,        /* 3M+9N step of the last try_step */
    BA_GET_CAMS = 7,      /* 15N  R(9 row-major), T(3), f, k1, k2 */
    BA_GET_POINTS = 8,    /* 3M */
    BA_GET_CAMS_TEST = 9,
    BA_GET_POINTS_TEST = 10
} ba_get_what;
int ba_solver_get(ba_solver *s, int what, double *out, size_t n);
/* Keep a copy of the reduced camera matrix / rhs of each try_step before it is factored in place, so that
 * BA_GET_S / BA_GET_RHS can return it (parity tests; costs one D x D device copy per trial). */
int ba_solver_keep_intermediates(ba_solver *s, int on);
/* Overwrite the resident parameters x (cam15: 15N, pts: 3M of this shard, file order). */
int ba_solver_set_state(ba_solver *s, const double *cam15, const double *pts);

/* Per-phase device time (ms, HIP events on the solver's stream) accumulated since the last reset. */
typedef struct {
    double linearize_ms;  /* residual + Jacobian + gradient / J_c^T J_c, per outer iteration */
    double eliminate_ms;  /* per-point elimination (3x3 LDL^T or Householder QR) */
    double schur_ms;      /* reduced camera matrix assembly */
    double factor_ms;     /* dense LDL^T + triangular solves */
    double backsub_ms;    /* point back-substitution + retraction */
    double test_eval_ms;  /* residual at xTest + scalar reductions */
    double comm_ms;       /* device time of the all-reduces (HIP events around them on the solver's stream) */
    double trial_ms;      /* whole trial, device time (the only per-trial figure when the trial is replayed as hipGraphs) */
    long long n_linearize, n_trials, n_graph_trials;
} ba_timing;
int ba_solver_timing(ba_solver *s, ba_timing *out, int reset);

/* Bench hooks: replay one phase `reps` times on the solver's stream and return the mean device ms per launch
 * (HIP events on that stream).  phase: 0 residual eval, 1 residual+Jacobian, 2 point elimination,
 * 3 Schur assembly, 4 Schur assembly + dense factor + solve, 5 back-substitution + retraction,
 * 6 dense factorisation only, 7 backward sweep only (6 / 7 rebuild S untimed before every repetition). */
int ba_solver_time_phase(ba_solver *s, int phase, int reps, double lambda, double *ms_per_launch);

/* A hand-off between workgroups of one launch that times out (the fused factorisation's row flag, the one-launch back sweep's
 * sentinel: waits bounded by design) does not end ba_minimize: the trial is repeated -- and the run continued -- with one launch per
 * step, where nobody waits for anybody; a second failure returns BA_ERR_HIP.  ba_solver_recoveries counts such repeats.
 * ba_minimize gives up with BA_ERR_HIP when no LM row appears for BA_WATCHDOG_S seconds (default 600): it does NOT wait for the
 * stream then, the handle is dead (every later call returns BA_ERR_HIP, ba_solver_free releases no device memory) and a retry
 * belongs in a fresh process.  BA_ERR_COMM: the shards of a sharded solve took different accept / stop decisions (guard slot of the
 * scalar all-reduce). */
int ba_solver_recoveries(const ba_solver *s);

/* Test hook for the failure paths of the in-launch hand-offs (no reference counterpart).  which = 2: arms a fault -- the row
 * workgroups of the fused factorisation stay silent and the panel's wait is short -- so that the next ba_minimize meets the
 * time-out on its first trial and has to recover (returns BA_ERR_ARG when the reduced system is a single block column).
 * which = 3: arms a 4-second kernel in front of the next LM trial (with BA_WATCHDOG_S = 1 ba_minimize must give up).  which = 1: runs the one-launch
 * backward sweep with the workgroup at the head of its dependency chain missing and a short spin bound, so the others
 * wait for unknowns that are never published.  Returns what the production path returns for that: BA_ERR_HIP (the kernels
 * raise a device error word that is read back with the trial's scalars); BA_ERR_ARG when the reduced system has fewer than
 * four 64-wide block columns (a single group has nobody to wait for). */
int ba_solver_selftest(ba_solver *s, int which);

/* Library / device info: fills name (<= n bytes), returns the number of CUs via *cus. */
int ba_device_info(int device, char *name, size_t n, int *cus);
const char *ba_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BA_MI355X_H */
