/*
 * ba_referee.c -- TEST INFRASTRUCTURE ONLY: the CPU oracle instantiated a third time with S = __float128.
 *
 * Why: the reference ships no golden output and cannot be built here, so the fp64 oracle is
 * "parity unpinned", and in the ill-conditioned part of an LM run (lambda at its 1e-10 floor, cond(J'J + lambda I)
 * ~ 1e20) two correct fp64 solvers disagree about the step by O(1).  The referee decides who is closer to the truth:
 * it evaluates ONE trial -- energy at x, step of (J'J + lambda I) dx = -J'r by the solver symbol's own elimination,
 * retraction, test energy, rho denominator -- in 113-bit arithmetic from a state (x, lambda) given in double.  The
 * fp64 oracle and the GPU are then both measured against it (tests/test_gpu_referee.py, tests/golden/referee_*.json).
 *
 * Same source as the oracle (ba_oracle_impl.h, with its reference file:line citations), so it inherits any misreading
 * of the reference's FORMULAS; what it removes is rounding as an excuse.
 *
 * Build: make -C oracle   ->  oracle/libba_referee.so   (gcc, -lquadmath)
 */
#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define S __float128
#define FN(name) CAT(name, _f128)
#define MINNORMAL 3.3621031431120935062626778173217526e-4932Q /* FLT128_MIN */
#define SQRT sqrtq
#define FABS fabsq
#define SIN sinq
#define COS cosq
#define POW powq
#include "ba_oracle_impl.h"

/* One LM trial in quad precision from a double state.
 *   kind: 0 QRKIT, 1 QRCHOL, 2 CHOLESKY, 3 MOREQR (ba_oracle_impl.h: ora_step)
 *   cam15 (15N), pts (3M), meas (2K), lambda: doubles, converted exactly
 *   out[8] (rounded to double): 0 energy at x, 1 test energy at x (+) dx, 2 rho denominator dx'(lambda dx - J'r... see :375),
 *                               3 |dx|, 4 max diag(J'J), 5 |J'r|, 6 backward error |(J'J + lambda I) dx + J'r| / |J'r|, 7 spare
 *   dx_out (3M + 9N doubles, may be NULL): the quad step rounded to double
 * Returns 0, or the oracle's error code. */
int ref_trial(int kind, int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas, double tau,
              const double *cam15, const double *pts, double lambda, double *out, double *dx_out)
{
    const size_t np = 3 * (size_t)M + 9 * (size_t)N;
    S *c = (S *)malloc(sizeof(S) * 15 * (size_t)N), *p = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    S *ms = (S *)malloc(sizeof(S) * 2 * (size_t)K), *f = (S *)malloc(sizeof(S) * 2 * (size_t)K);
    S *Jc = (S *)malloc(sizeof(S) * 18 * (size_t)K), *Jp = (S *)malloc(sizeof(S) * 6 * (size_t)K);
    S *dx = (S *)calloc(np, sizeof(S)), *g = (S *)malloc(sizeof(S) * np);
    S *ct = (S *)malloc(sizeof(S) * 15 * (size_t)N), *pt = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    if (!c || !p || !ms || !f || !Jc || !Jp || !dx || !g || !ct || !pt) return -1;
    for (size_t i = 0; i < 15 * (size_t)N; i++) c[i] = cam15[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) p[i] = pts[i];
    for (size_t i = 0; i < 2 * (size_t)K; i++) ms[i] = meas[i];
    const S e = ora_residuals_f128(N, M, K, c, p, cam_idx, pt_idx, ms, (S)tau, f);
    ora_jacobian_f128(N, M, K, c, p, cam_idx, pt_idx, ms, (S)tau, Jc, Jp);
    S dmax = 0;
    int rc = ora_step_f128(kind, N, M, K, cam_idx, pt_idx, Jc, Jp, f, (S)lambda, dx, NULL, NULL, g, &dmax);
    if (!rc) {
        ora_retract_f128(N, M, c, p, dx, ct, pt);
        const S et = ora_residuals_f128(N, M, K, ct, pt, cam_idx, pt_idx, ms, (S)tau, NULL);
        S rs = 0, dn = 0, gn = 0;
        for (size_t i = 0; i < np; i++) {
            rs += dx[i] * ((S)lambda * dx[i] + g[i]); /* BacktrackLevMarqQRChol.h:375 (g = JtRes = -J'r) */
            dn += dx[i] * dx[i];
            gn += g[i] * g[i];
        }
        /* backward error in the normal equations, with the quad Jacobian */
        S *res = (S *)calloc(np, sizeof(S));
        for (int i = 0; i < K; i++) {
            const S *A = Jc + 18 * (size_t)i, *B = Jp + 6 * (size_t)i;
            const S *dc = dx + 3 * (size_t)M + 9 * (size_t)cam_idx[i], *dp = dx + 3 * (size_t)pt_idx[i];
            S j0 = 0, j1 = 0;
            for (int q = 0; q < 9; q++) { j0 += A[q] * dc[q]; j1 += A[9 + q] * dc[q]; }
            for (int q = 0; q < 3; q++) { j0 += B[q] * dp[q]; j1 += B[3 + q] * dp[q]; }
            S *rc_ = res + 3 * (size_t)M + 9 * (size_t)cam_idx[i], *rp = res + 3 * (size_t)pt_idx[i];
            for (int q = 0; q < 9; q++) rc_[q] += A[q] * j0 + A[9 + q] * j1;
            for (int q = 0; q < 3; q++) rp[q] += B[q] * j0 + B[3 + q] * j1;
        }
        S rn = 0;
        for (size_t i = 0; i < np; i++) {
            const S v = res[i] + (S)lambda * dx[i] - g[i];
            rn += v * v;
        }
        free(res);
        out[0] = (double)e; out[1] = (double)et; out[2] = (double)rs; out[3] = (double)sqrtq(dn); out[4] = (double)dmax;
        out[5] = (double)sqrtq(gn); out[6] = (double)(sqrtq(rn) / sqrtq(gn)); out[7] = 0;
        if (dx_out)
            for (size_t i = 0; i < np; i++) dx_out[i] = (double)dx[i];
    }
    free(c); free(p); free(ms); free(f); free(Jc); free(Jp); free(dx); free(g); free(ct); free(pt);
    return rc;
}

/* Elimination + reduced camera system of one trial in quad precision (no factorisation): S (D x D column-major, full
 * symmetric) and the reduced rhs, rounded to double.  Decides whose S is closer to the truth where fp64 assemblies differ by
 * more than the 1e-11 the small problems show (ill-conditioned 3x3 point blocks under a small lambda). */
int ref_reduced(int kind, int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas, double tau,
                const double *cam15, const double *pts, double lambda, double *S_out, double *rhs_out)
{
    const size_t np = 3 * (size_t)M + 9 * (size_t)N, D = 9 * (size_t)N;
    S *c = (S *)malloc(sizeof(S) * 15 * (size_t)N), *p = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    S *ms = (S *)malloc(sizeof(S) * 2 * (size_t)K), *f = (S *)malloc(sizeof(S) * 2 * (size_t)K);
    S *Jc = (S *)malloc(sizeof(S) * 18 * (size_t)K), *Jp = (S *)malloc(sizeof(S) * 6 * (size_t)K);
    S *dx = (S *)calloc(np, sizeof(S)), *Sq = (S *)malloc(sizeof(S) * D * D), *rq = (S *)malloc(sizeof(S) * D);
    if (!c || !p || !ms || !f || !Jc || !Jp || !dx || !Sq || !rq) return -1;
    for (size_t i = 0; i < 15 * (size_t)N; i++) c[i] = cam15[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) p[i] = pts[i];
    for (size_t i = 0; i < 2 * (size_t)K; i++) ms[i] = meas[i];
    (void)ora_residuals_f128(N, M, K, c, p, cam_idx, pt_idx, ms, (S)tau, f);
    ora_jacobian_f128(N, M, K, c, p, cam_idx, pt_idx, ms, (S)tau, Jc, Jp);
    const int rc = ora_step_f128(kind | 256, N, M, K, cam_idx, pt_idx, Jc, Jp, f, (S)lambda, dx, Sq, rq, NULL, NULL);
    if (!rc) {
        for (size_t i = 0; i < D * D; i++) S_out[i] = (double)Sq[i];
        for (size_t i = 0; i < D; i++) rhs_out[i] = (double)rq[i];
    }
    free(c); free(p); free(ms); free(f); free(Jc); free(Jp); free(dx); free(Sq); free(rq);
    return rc;
}

/* Energy only (quad) of a double state: what an fp64 evaluation of the test energy should have returned. */
double ref_energy(int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas, double tau,
                  const double *cam15, const double *pts)
{
    S *c = (S *)malloc(sizeof(S) * 15 * (size_t)N), *p = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    S *ms = (S *)malloc(sizeof(S) * 2 * (size_t)K);
    for (size_t i = 0; i < 15 * (size_t)N; i++) c[i] = cam15[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) p[i] = pts[i];
    for (size_t i = 0; i < 2 * (size_t)K; i++) ms[i] = meas[i];
    const S e = ora_residuals_f128(N, M, K, c, p, cam_idx, pt_idx, ms, (S)tau, NULL);
    free(c); free(p); free(ms);
    return (double)e;
}

/* The Jacobian as the DERIVATIVE OF THE RESIDUAL FUNCTION, not as a restatement of the reference's hand-written dE_pos: central
 * differences of ora_residuals through the reference's update_params (BAFunctor.h:299-342: additive, left-multiplication by the
 * Rodrigues matrix of the increment for the rotation), in quad precision with one Richardson step -- (4 D(h/2) - D(h)) / 3,
 * truncation O(h^4), rounding 1e-34 / h.  Pins ora_jacobian (and through it k_eval's Jacobian) to BAFunctor::E_pos independently of
 * BAFunctor::dE_pos:181-297; the double finite differences of tests/test_oracle_math.py could only do that to 2e-3.
 * The rotation increment uses the exponential map WITHOUT the reference's 1e-6 cut-off (MathUtils.h:74 leaves R untouched below
 * it -- a property of tiny steps, not of the derivative): a pixel moves ~1e4 px per radian and the robust kernel bends on the
 * scale tau = 0.5 px, so steps above the cut-off are far outside the linear range (measured: 9e-2 relative error at h = 2e-5).
 * Steps: 1e-9 (rotation), 1e-8 x scale (others).  Jc_out: K x 18 (2 x 9 row-major), Jp_out: K x 6 -- ora_jacobian's layout. */
static void ref_expmap(const S *om, S *R)
{
    const S t2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2], theta = sqrtq(t2);
    const S J[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    const S c1 = theta > 0 ? sinq(theta) / theta : (S)1, c2 = theta > 0 ? ((S)1 - cosq(theta)) / t2 : (S)0.5;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            S a = 0;
            for (int k = 0; k < 3; k++) a += J[i * 3 + k] * J[k * 3 + j];
            R[i * 3 + j] = (i == j ? (S)1 : (S)0) + c1 * J[i * 3 + j] + c2 * a;
        }
}

int ref_jacobian_fd(int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas, double tau,
                    const double *cam15, const double *pts, double *Jc_out, double *Jp_out)
{
    (void)N; (void)M;
    const int zero = 0;
    for (int i = 0; i < K; i++) {
        S cam[15], X[3], ms[2];
        for (int k = 0; k < 15; k++) cam[k] = cam15[15 * (size_t)cam_idx[i] + k];
        for (int k = 0; k < 3; k++) X[k] = pts[3 * (size_t)pt_idx[i] + k];
        ms[0] = meas[2 * (size_t)i]; ms[1] = meas[2 * (size_t)i + 1];
        for (int c = 0; c < 12; c++) { /* camera column c < 9: [T(3) omega(3) f k1 k2]; 9..11: the point */
            S h = (c >= 3 && c < 6) ? (S)1e-9 : (S)1e-8;
            if (c == 6) h *= 1000; /* focal length ~ 1e3 */
            S D[2][2];
            for (int lev = 0; lev < 2; lev++) {
                const S hh = lev ? h / 2 : h;
                S f2[2][2];
                for (int sgn = 0; sgn < 2; sgn++) {
                    const S d = sgn ? -hh : hh;
                    S co[15], Xo[3] = {X[0], X[1], X[2]};
                    for (int k = 0; k < 15; k++) co[k] = cam[k];
                    if (c < 3) co[9 + c] += d;
                    else if (c < 6) {
                        S om[3] = {0, 0, 0}, dR[9];
                        om[c - 3] = d;
                        ref_expmap(om, dR);
                        for (int r = 0; r < 3; r++)
                            for (int q = 0; q < 3; q++) {
                                S a = 0;
                                for (int k = 0; k < 3; k++) a += dR[r * 3 + k] * cam[k * 3 + q];
                                co[r * 3 + q] = a;
                            }
                    } else if (c < 9) co[12 + (c - 6)] += d;
                    else Xo[c - 9] += d;
                    (void)ora_residuals_f128(1, 1, 1, co, Xo, &zero, &zero, ms, (S)tau, f2[sgn]);
                }
                D[lev][0] = (f2[0][0] - f2[1][0]) / (2 * hh);
                D[lev][1] = (f2[0][1] - f2[1][1]) / (2 * hh);
            }
            for (int r = 0; r < 2; r++) {
                const S d = (4 * D[1][r] - D[0][r]) / 3;
                if (c < 9) Jc_out[18 * (size_t)i + 9 * r + c] = (double)d;
                else Jp_out[6 * (size_t)i + 3 * r + (c - 9)] = (double)d;
            }
        }
    }
    return 0;
}


/* The whole LM loop (ora_minimize, ba_oracle_impl.h) in quad precision from a double start: the trajectory exact arithmetic would
 * follow, against which the two fp64 sides' FINAL energies are measured (tests/golden/make_referee.py: free runs).  meas, cam15,
 * pts, lm as in ora_minimize_f64; cam15 / pts return the final state rounded to double; trace: max_trials x 8 doubles. */
int ref_minimize(int kind, int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas, double tau, double *cam15,
                 double *pts, const double *lm, int max_iter, int max_fun_ev, int max_trials, double *trace, int *ntrials_out)
{
    S *c = (S *)malloc(sizeof(S) * 15 * (size_t)N), *p = (S *)malloc(sizeof(S) * 3 * (size_t)M), *ms = (S *)malloc(sizeof(S) * 2 * (size_t)K);
    if (!c || !p || !ms) return -4;
    for (size_t i = 0; i < 15 * (size_t)N; i++) c[i] = cam15[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) p[i] = pts[i];
    for (size_t i = 0; i < 2 * (size_t)K; i++) ms[i] = meas[i];
    const int status = ora_minimize_f128(kind, N, M, K, cam_idx, pt_idx, ms, (S)tau, c, p, lm, max_iter, max_fun_ev, max_trials, trace, ntrials_out, NULL);
    for (size_t i = 0; i < 15 * (size_t)N; i++) cam15[i] = (double)c[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) pts[i] = (double)p[i];
    free(c); free(p); free(ms);
    return status;
}

/* ---- the oracle a fourth time: S = long double (x87 extended: 64-bit significand, eps 1.1e-19 = fp64's / 2048) -----------------------
 * Round 4: a free run in this arithmetic costs ~3x an fp64 one (a __float128 run: ~1000x), so WHOLE ENSEMBLES of free runs are affordable.
 * They answer what single quad runs cannot: does the DISTRIBUTION of final energies of the reference algorithm depend on the size of the
 * rounding noise of its linear solves?  (tests/golden/make_referee.py: ensemble_*_x87) */
#undef S
#undef FN
#undef MINNORMAL
#undef SQRT
#undef FABS
#undef SIN
#undef COS
#undef POW
#define S long double
#define FN(name) CAT(name, _f80)
#define MINNORMAL 3.36210314311209350626e-4932L /* LDBL_MIN */
#define SQRT sqrtl
#define FABS fabsl
#define SIN sinl
#define COS cosl
#define POW powl
#include "ba_oracle_impl.h"

int ref80_minimize(int kind, int N, int M, int K, const int *cam_idx, const int *pt_idx, const double *meas, double tau, double *cam15,
                   double *pts, const double *lm, int max_iter, int max_fun_ev, int max_trials, double *trace, int *ntrials_out)
{
    S *c = (S *)malloc(sizeof(S) * 15 * (size_t)N), *p = (S *)malloc(sizeof(S) * 3 * (size_t)M), *ms = (S *)malloc(sizeof(S) * 2 * (size_t)K);
    if (!c || !p || !ms) return -4;
    for (size_t i = 0; i < 15 * (size_t)N; i++) c[i] = cam15[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) p[i] = pts[i];
    for (size_t i = 0; i < 2 * (size_t)K; i++) ms[i] = meas[i];
    const int status = ora_minimize_f80(kind, N, M, K, cam_idx, pt_idx, ms, (S)tau, c, p, lm, max_iter, max_fun_ev, max_trials, trace, ntrials_out, NULL);
    for (size_t i = 0; i < 15 * (size_t)N; i++) cam15[i] = (double)c[i];
    for (size_t i = 0; i < 3 * (size_t)M; i++) pts[i] = (double)p[i];
    free(c); free(p); free(ms);
    return status;
}
