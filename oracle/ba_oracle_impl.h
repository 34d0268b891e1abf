/*
 * ba_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle), never the product path.
 *
 * Plain-C restatement of the Levenberg-Marquardt bundle-adjustment hot path of
 * jasvob/BundleAdjustment_Benchmarks.  Included twice by ba_oracle.c, once with
 * S = double (suffix _f64) and once with S = float (suffix _f32), mirroring the
 * reference's `typedef double Scalar;` switch (src/BATypeUtils.h:6-7).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or recorded
 * output, and cannot be compiled here (private Eigen/QRKit fork + SuiteSparse
 * absent).  This oracle is pinned only by (1) the two BAL data
 * files the reference ships, (2) finite-difference / mpmath checks of the
 * Jacobian, (3) an independent scipy sparse solve of (J'J + lambda I) dx = -J'r
 * (tests/test_oracle.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this file.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */

/* ---- per-observation geometry --------------------------------------------------------- */

/* Camera state layout (15 scalars): R row-major [0..8], T [9..11], f=K(0,0) [12], k1 [13], k2 [14].
 * The reference holds K,R,T,K^-1,R^T,centre per camera (CameraMatrix.h:72-76); only these 15 are state. */

/* Math::createRotationMatrixRodrigues, MathUtils.h:66-82 (theta <= 1e-6 -> identity). */
static void FN(rodrigues)(const S *om, S *R)
{
    const S theta = SQRT(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    if (FABS(theta) > (S)1e-6) {
        /* J = [om]x (MathUtils.h:13-21), J2 = J*J */
        S J[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
        S J2[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                S a = 0;
                for (int k = 0; k < 3; k++) a += J[i * 3 + k] * J[k * 3 + j];
                J2[i * 3 + j] = a;
            }
        const S c1 = SIN(theta) / theta;
        const S c2 = ((S)1.0 - COS(theta)) / (theta * theta);
        for (int i = 0; i < 9; i++) R[i] = R[i] + c1 * J[i] + c2 * J2[i];
    }
}

/* bundle_adjustment_large.cpp:81-99: K00=K11=-f/avg_focal (avg_focal=1), R=Rodrigues(om),
 * distortion (k1 f^2, k2 f^4). cams9 = [om(3), T(3), f, k1, k2] per camera as in the BAL file. */
void FN(ora_init_cams)(int N, const double *cams9, S *cam15)
{
    for (int i = 0; i < N; i++) {
        const double *c = cams9 + 9 * (size_t)i;
        S *o = cam15 + 15 * (size_t)i;
        S om[3] = {(S)c[0], (S)c[1], (S)c[2]};
        FN(rodrigues)(om, o);
        o[9] = (S)c[3]; o[10] = (S)c[4]; o[11] = (S)c[5];
        const S f = (S)c[6], k1 = (S)c[7], k2 = (S)c[8];
        const S f2 = f * f;
        o[12] = -f / (S)1.0;
        o[13] = k1 * f2;
        o[14] = k2 * f2 * f2;
    }
}

/* BAFunctor::psi / psi_weight, BAFunctor.h:147-148 */
static inline S FN(psi)(S tau2, S r2) { return (r2 < tau2) ? r2 * ((S)2.0 - r2 / tau2) / (S)4.0 : tau2 / (S)4.0; }
static inline S FN(psi_weight)(S tau2, S r2) { S w = (S)1.0 - r2 / tau2; return w > (S)0.0 ? w : (S)0.0; }

#define EPS_PSI ((S)1e-15) /* BAFunctor.h:159 */

/* BAFunctor::projectPoint, BAFunctor.h:151-156 (+ CameraMatrix.cpp:259-261, DistortionFunction.cpp:14-23) */
static inline void FN(project)(const S *cam, const S *X, S *q)
{
    const S XX0 = cam[0] * X[0] + cam[1] * X[1] + cam[2] * X[2] + cam[9];
    const S XX1 = cam[3] * X[0] + cam[4] * X[1] + cam[5] * X[2] + cam[10];
    const S XX2 = cam[6] * X[0] + cam[7] * X[1] + cam[8] * X[2] + cam[11];
    const S xu0 = XX0 / XX2, xu1 = XX1 / XX2;
    const S r2 = xu0 * xu0 + xu1 * xu1;
    const S r4 = r2 * r2;
    const S kr = 1 + cam[13] * r2 + cam[14] * r4;
    q[0] = cam[12] * (kr * xu0);
    q[1] = cam[12] * (kr * xu1);
}

/* BAFunctor::E_pos, BAFunctor.h:160-178.  fvec is obs-major interleaved (2i, 2i+1).
 * Returns the energy fvec.squaredNorm() (BacktrackLevMarqQRChol.h:261). */
S FN(ora_residuals)(int N, int M, int K, const S *cam15, const S *pts, const int *cam_idx, const int *pt_idx,
                    const S *meas, S tau, S *fvec)
{
    (void)N; (void)M;
    const S tau2 = tau * tau;
    S energy = 0;
    for (int i = 0; i < K; i++) {
        S q[2];
        FN(project)(cam15 + 15 * (size_t)cam_idx[i], pts + 3 * (size_t)pt_idx[i], q);
        const S r0 = q[0] - meas[2 * (size_t)i], r1 = q[1] - meas[2 * (size_t)i + 1];
        const S r2 = r0 * r0 + r1 * r1;
        const S sqrt_psi = SQRT(FN(psi)(tau2, r2));
        const S nr = SQRT(r2);
        const S rnorm_r = (S)1.0 / (EPS_PSI > nr ? EPS_PSI : nr);
        const S e0 = r0 * sqrt_psi * rnorm_r, e1 = r1 * sqrt_psi * rnorm_r;
        if (fvec) { fvec[2 * (size_t)i] = e0; fvec[2 * (size_t)i + 1] = e1; }
        energy += e0 * e0 + e1 * e1;
    }
    return energy;
}

/* BAFunctor::dE_pos, BAFunctor.h:181-297 (+ poseDerivatives :126-142, DistortionFunction.cpp:25-51).
 * Jc: K blocks of 2x9 row-major, camera columns ordered [T(3), omega(3), f, k1, k2] (:186-191,265-284);
 * Jp: K blocks of 2x3 row-major (:287-292). */
void FN(ora_jacobian)(int N, int M, int K, const S *cam15, const S *pts, const int *cam_idx, const int *pt_idx,
                      const S *meas, S tau, S *Jc, S *Jp)
{
    (void)N; (void)M;
    const S tau2 = tau * tau;
    for (int i = 0; i < K; i++) {
        const S *cam = cam15 + 15 * (size_t)cam_idx[i];
        const S *X = pts + 3 * (size_t)pt_idx[i];
        /* poseDerivatives: XX = R X + T ; d_dRT = [I | -[XX - T]x] ; d_dX = R */
        const S RX0 = cam[0] * X[0] + cam[1] * X[1] + cam[2] * X[2];
        const S RX1 = cam[3] * X[0] + cam[4] * X[1] + cam[5] * X[2];
        const S RX2 = cam[6] * X[0] + cam[7] * X[1] + cam[8] * X[2];
        const S XX0 = RX0 + cam[9], XX1 = RX1 + cam[10], XX2 = RX2 + cam[11];
        const S v0 = XX0 - cam[9], v1 = XX1 - cam[10], v2 = XX2 - cam[11];
        /* -[v]x */
        const S mJ[9] = {0, v2, -v1, -v2, 0, v0, v1, -v0, 0};
        const S xu0 = XX0 / XX2, xu1 = XX1 / XX2;
        const S r2u = xu0 * xu0 + xu1 * xu1, r4u = r2u * r2u;
        const S k1 = cam[13], k2 = cam[14], f = cam[12];
        const S kr = 1 + k1 * r2u + k2 * r4u;
        const S xd0 = kr * xu0, xd1 = kr * xu1;
        /* dxu_dXX (:219-221) */
        const S a00 = (S)1.0 / XX2, a02 = -XX0 / (XX2 * XX2);
        const S a11 = (S)1.0 / XX2, a12 = -XX1 / (XX2 * XX2);
        /* dxd_dxu (DistortionFunction.cpp:38-51) */
        const S dkr = 2 * k1 + 4 * k2 * r2u;
        const S d00 = kr + xu0 * xu0 * dkr, d01 = xu0 * xu1 * dkr, d11 = kr + xu1 * xu1 * dkr;
        /* dp_dxu = diag(f,f) * dxd_dxu ; dp_dXX = dp_dxu * dxu_dXX (:223-225) */
        const S p00 = f * d00, p01 = f * d01, p10 = f * d01, p11 = f * d11;
        S dpX[6]; /* 2x3 */
        dpX[0] = p00 * a00; dpX[1] = p01 * a11; dpX[2] = p00 * a02 + p01 * a12;
        dpX[3] = p10 * a00; dpX[4] = p11 * a11; dpX[5] = p10 * a02 + p11 * a12;
        /* outer derivative of the psi residual (:227-242) */
        const S q0 = f * xd0, q1 = f * xd1;
        const S r0 = q0 - meas[2 * (size_t)i], r1 = q1 - meas[2 * (size_t)i + 1];
        const S r2 = r0 * r0 + r1 * r1;
        const S W = FN(psi_weight)(tau2, r2);
        const S sqrt_psi = SQRT(FN(psi)(tau2, r2));
        const S rsqrt_psi = (S)1.0 / (EPS_PSI > sqrt_psi ? EPS_PSI : sqrt_psi);
        const S rcp_r2 = (S)1.0 / (EPS_PSI > r2 ? EPS_PSI : r2);
        const S nr = SQRT(r2);
        const S rnorm_r = (S)1.0 / (EPS_PSI > nr ? EPS_PSI : nr);
        const S rr00 = r0 * r0 * rnorm_r, rr01 = r0 * r1 * rnorm_r, rr11 = r1 * r1 * rnorm_r;
        const S c1 = W / (S)2.0 * rsqrt_psi, c2 = sqrt_psi * rcp_r2;
        const S o00 = c1 * rr00 + c2 * (nr - rr00);
        const S o01 = c1 * rr01 + c2 * ((S)0 - rr01);
        const S o11 = c1 * rr11 + c2 * (nr - rr11);
        /* Jblock (2x12): [0..5]=dp_dXX*d_dRT, [6]=xd, [7,8]=f*d xd/d(k1,k2), [9..11]=dp_dXX*R (:244-258) */
        S Jb[24];
        for (int r = 0; r < 2; r++) {
            const S *d = dpX + 3 * r;
            S *o = Jb + 12 * r;
            o[0] = d[0]; o[1] = d[1]; o[2] = d[2];
            for (int c = 0; c < 3; c++) o[3 + c] = d[0] * mJ[c] + d[1] * mJ[3 + c] + d[2] * mJ[6 + c];
            for (int c = 0; c < 3; c++) o[9 + c] = d[0] * cam[c] + d[1] * cam[3 + c] + d[2] * cam[6 + c];
        }
        Jb[6] = xd0; Jb[12 + 6] = xd1;
        Jb[7] = f * (xu0 * r2u); Jb[8] = f * (xu0 * r4u);
        Jb[12 + 7] = f * (xu1 * r2u); Jb[12 + 8] = f * (xu1 * r4u);
        /* Jblock = outer_deriv * Jblock (:261) */
        S *jc = Jc + 18 * (size_t)i, *jp = Jp + 6 * (size_t)i;
        for (int c = 0; c < 12; c++) {
            const S t0 = o00 * Jb[c] + o01 * Jb[12 + c];
            const S t1 = o01 * Jb[c] + o11 * Jb[12 + c];
            if (c < 9) { jc[c] = t0; jc[9 + c] = t1; }
            else { jp[c - 9] = t0; jp[3 + c - 9] = t1; }
        }
    }
}

/* BAFunctor::update_params, BAFunctor.h:299-342.  dx layout = Jacobian column order:
 * 3M point coordinates first, then per camera [T(3), omega(3), f, k1, k2]. */
void FN(ora_retract)(int N, int M, const S *cam_in, const S *pts_in, const S *dx, S *cam_out, S *pts_out)
{
    const S *dc = dx + 3 * (size_t)M;
    for (int i = 0; i < N; i++) {
        const S *c = cam_in + 15 * (size_t)i;
        S *o = cam_out + 15 * (size_t)i;
        const S *p = dc + 9 * (size_t)i;
        S dR[9], Rn[9];
        FN(rodrigues)(p + 3, dR);
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                S a = 0;
                for (int k = 0; k < 3; k++) a += dR[r * 3 + k] * c[k * 3 + q];
                Rn[r * 3 + q] = a;
            }
        for (int k = 0; k < 9; k++) o[k] = Rn[k];
        o[9] = c[9] + p[0]; o[10] = c[10] + p[1]; o[11] = c[11] + p[2];
        o[12] = c[12] + p[6];
        o[13] = c[13] + p[7];
        o[14] = c[14] + p[8];
    }
    for (size_t i = 0; i < 3 * (size_t)M; i++) pts_out[i] = pts_in[i] + dx[i];
}

/* Utils::showErrorStatistics + showObjective, Utils.h:10-68 (CameraMatrix::projectPoint(dist,X),
 * CameraMatrix.cpp:225-236).  out = {mean reprojection error, inlier mean error, nInliers, objective}.
 * Quirk kept: showObjective feeds the NORM (not squared) into Utils::psi (Utils.h:61-62). */
void FN(ora_stats)(int N, int M, int K, const S *cam15, const S *pts, const int *cam_idx, const int *pt_idx,
                   const S *meas, S tau, double *out4)
{
    (void)N; (void)M;
    const S tau2 = tau * tau, tau4 = tau2 * tau2;
    S mean = 0, inl = 0, obj = 0;
    int nin = 0;
    for (int k = 0; k < K; k++) {
        S q[2];
        FN(project)(cam15 + 15 * (size_t)cam_idx[k], pts + 3 * (size_t)pt_idx[k], q);
        const S d0 = q[0] - meas[2 * (size_t)k], d1 = q[1] - meas[2 * (size_t)k + 1];
        const S err = SQRT(d0 * d0 + d1 * d1);
        mean += err;
        if (err <= tau) { nin++; inl += err; }
        const S r2 = err; /* sic */
        const S r4 = r2 * r2;
        obj += (r2 < tau2) ? r2 * ((S)3.0 - (S)3.0 * r2 / tau2 + r4 / tau4) / (S)6.0 : tau2 / (S)6.0;
    }
    out4[0] = (double)(mean / K);
    out4[1] = (double)(inl / nin);
    out4[2] = (double)nin;
    out4[3] = (double)obj;
}

/* ---- linear algebra of one LM trial --------------------------------------------------- */

/* Dense LDL^T, lower triangle, column-major n x n with leading dimension n, in place:
 * strictly-lower part <- L (unit diagonal implied), diagonal <- D.  Stands in for
 * Eigen::SimplicialLDLT on the (block-dense) reduced camera matrix
 * (BAFunctor.h:106, BacktrackLevMarqQRChol.h:339; BacktrackLevMarqCholesky.h:156,278).
 * No pivoting, no sqrt: survives small negative pivots like SimplicialLDLT does. */
static void FN(dense_ldlt)(int n, S *A)
{
    /* Blocked right-looking form (64-wide panels; the trailing update is the only loop OpenMP splits, by columns).  Every
     * element still receives its updates in ascending pivot order with the products formed as L(i,k) * (L(j,k) D(k)), so the
     * result is bit-identical to the plain left-looking column loop this replaced (and to itself for any thread count):
     * the committed fixtures of the oracle's trajectories stay valid.  It is here for the CPU baseline's sake -- the
     * unblocked loop re-read the whole factor for every column (1.3 LM it/s at config 4). */
    enum { PB = 64 };
    S *w = (S *)malloc(sizeof(S) * (size_t)PB * (size_t)n); /* w[(k - j0) * n + c] = L(c,k) D(k) */
    for (int j0 = 0; j0 < n; j0 += PB) {
        const int j1 = j0 + PB < n ? j0 + PB : n;
        /* panel: columns j0..j1-1, updates by the panel's own earlier columns only (earlier panels have been applied) */
        for (int j = j0; j < j1; j++) {
            S *col = A + (size_t)j * n;
            S d = col[j];
            for (int k = j0; k < j; k++) {
                const S wk = A[(size_t)k * n + j] * A[(size_t)k * n + k];
                w[(size_t)(k - j0) * n + j] = wk;
                d -= A[(size_t)k * n + j] * wk;
            }
            col[j] = d;
            for (int k = j0; k < j; k++) {
                const S wk = w[(size_t)(k - j0) * n + j];
                const S *ck = A + (size_t)k * n;
                for (int i = j + 1; i < n; i++) col[i] -= ck[i] * wk;
            }
            const S inv = (S)1.0 / d;
            for (int i = j + 1; i < n; i++) col[i] *= inv;
        }
        /* trailing update of the columns behind the panel */
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 8) if (n - j1 > 256)
#endif
        for (int c = j1; c < n; c++) {
            S *col = A + (size_t)c * n;
            S d = col[c];
            for (int k = j0; k < j1; k++) {
                const S *ck = A + (size_t)k * n;
                const S wk = ck[c] * ck[k];
                d -= ck[c] * wk;
                for (int i = c + 1; i < n; i++) col[i] -= ck[i] * wk;
            }
            col[c] = d;
        }
    }
    free(w);
}

static void FN(dense_ldlt_solve)(int n, const S *A, S *b)
{
    for (int j = 0; j < n; j++) {
        const S bj = b[j];
        const S *col = A + (size_t)j * n;
        for (int i = j + 1; i < n; i++) b[i] -= col[i] * bj;
    }
    for (int j = 0; j < n; j++) b[j] /= A[(size_t)j * n + j];
    for (int j = n - 1; j >= 0; j--) {
        const S *col = A + (size_t)j * n;
        S a = b[j];
        for (int i = j + 1; i < n; i++) a -= col[i] * b[i];
        b[j] = a;
    }
}

/* Dense Householder QR solve of min || A x - b ||, A is m x n column-major (ld m), overwritten.
 * Stands in for QRKit's DenseBlockedThinQR on the lower-right block (BAFunctor.h:101; README.md:14). */
static void FN(dense_qr_solve)(int m, int n, S *A, S *b, S *x)
{
    for (int j = 0; j < n; j++) {
        S *col = A + (size_t)j * m;
        S xn = 0;
        for (int i = j + 1; i < m; i++) xn += col[i] * col[i];
        const S alpha = col[j];
        if (xn == 0) continue; /* H = I */
        S beta = SQRT(alpha * alpha + xn);
        if (alpha > 0) beta = -beta;
        const S tau = (beta - alpha) / beta;
        const S sc = (S)1.0 / (alpha - beta);
        for (int i = j + 1; i < m; i++) col[i] *= sc;
        col[j] = beta;
        for (int c = j + 1; c <= n; c++) {
            S *cc = (c < n) ? A + (size_t)c * m : b;
            S w = cc[j];
            for (int i = j + 1; i < m; i++) w += col[i] * cc[i];
            w *= tau;
            cc[j] -= w;
            for (int i = j + 1; i < m; i++) cc[i] -= col[i] * w;
        }
    }
    for (int j = n - 1; j >= 0; j--) {
        S a = b[j];
        for (int c = j + 1; c < n; c++) a -= A[(size_t)c * m + j] * x[c];
        x[j] = a / A[(size_t)j * m + j];
    }
}

/* Householder QR of a TALL m x n matrix (column-major, ld m) with the right-hand side b riding along, by row chunks (a TSQR tree of
 * depth two): every chunk of CHK rows is factored by itself -- it stays in cache, the plain column loop above streams the whole
 * matrix from memory once per reflector -- and the chunks' R factors (+ the heads of their Q^T b) are stacked and factored again.
 * On exit R (n x n, upper triangle, column-major ld n) and c = the first n entries of Q^T b.  Used by MOREQR's two dense QRs
 * (BacktrackLevMarqMore.h:288, :328); the QRKIT / QRSPQR symbols keep the plain loop (their committed fixtures hold its bits). */
static int FN(dense_qr_factor_tsqr)(size_t m, int n, const S *A, const S *b, S *R, S *c)
{
    const size_t CHK = 2048;
    const size_t nch = (m + CHK - 1) / CHK;
    const size_t ms = nch * (size_t)n; /* rows of the stack */
    S *T = (S *)malloc(sizeof(S) * CHK * (size_t)(n + 1));
    S *St = (S *)calloc(ms * (size_t)(n + 1), sizeof(S));
    if (!T || !St) { free(T); free(St); return -1; }
    for (int pass = 0; pass < 2; pass++) {
        const size_t rows_total = pass == 0 ? m : ms, chunks = pass == 0 ? nch : 1;
        for (size_t g = 0; g < chunks; g++) {
            const size_t r0 = pass == 0 ? g * CHK : 0, mr = pass == 0 ? (r0 + CHK <= m ? CHK : m - r0) : ms;
            S *W = pass == 0 ? T : St; /* the stack is factored in place */
            const size_t ldw = pass == 0 ? mr : ms;
            if (pass == 0) {
                for (int j = 0; j < n; j++)
                    for (size_t i = 0; i < mr; i++) W[(size_t)j * ldw + i] = A[(size_t)j * m + r0 + i];
                for (size_t i = 0; i < mr; i++) W[(size_t)n * ldw + i] = b[r0 + i];
            }
            const int kmax = (size_t)n < mr ? n : (int)mr;
            for (int j = 0; j < kmax; j++) {
                S *col = W + (size_t)j * ldw;
                S xn = 0;
                for (size_t i = j + 1; i < mr; i++) xn += col[i] * col[i];
                const S alpha = col[j];
                /* H = I for a tail that is zero -- or whose squared norm is not a normal number: Eigen's makeHouseholder
                 * (Eigen/src/Householder/Householder.h, the library the reference's QR classes are built on; not vendored) sets
                 * tau = 0, beta = c0 when tailSqNorm <= numeric_limits<Scalar>::min().  A sum of denormal squares has only a few
                 * bits, and beta, tau formed from it would not fit v (what round 4 found in the GPU's kernels). */
                if (xn <= MINNORMAL) continue;
                S beta = SQRT(alpha * alpha + xn);
                if (alpha > 0) beta = -beta;
                const S tau = (beta - alpha) / beta;
                const S sc = (S)1.0 / (alpha - beta);
                for (size_t i = j + 1; i < mr; i++) col[i] *= sc;
                col[j] = beta;
                for (int cc_ = j + 1; cc_ <= n; cc_++) {
                    S *cc = W + (size_t)cc_ * ldw;
                    S w = cc[j];
                    for (size_t i = j + 1; i < mr; i++) w += col[i] * cc[i];
                    w *= tau;
                    cc[j] -= w;
                    for (size_t i = j + 1; i < mr; i++) cc[i] -= col[i] * w;
                }
            }
            if (pass == 0) { /* R and the head of Q^T b of this chunk into block g of the stack */
                for (int j = 0; j <= n; j++)
                    for (int i = 0; i < n && (size_t)i < mr; i++)
                        if (j == n || i <= j) St[(size_t)j * ms + g * (size_t)n + i] = W[(size_t)j * ldw + i];
            }
        }
        (void)rows_total;
    }
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) R[(size_t)j * n + i] = i <= j ? St[(size_t)j * ms + i] : (S)0;
    for (int i = 0; i < n; i++) c[i] = St[(size_t)n * ms + i];
    free(T); free(St);
    return 0;
}

/* Workspace of the per-point elimination shared by all three solver symbols:
 *   Z[i]    9x3 per observation (row-major): CHOLESKY  Z = A^T B L^-T     (W L^-T)
 *                                            QR        Z = R12_i^T = A^T Q1_i
 *   dinv[j] 3 per point:                     CHOLESKY  1/D of U_j+lambda I = L D L^T ; QR: 1
 *   t[j]    3 per point:                     CHOLESKY  L^-1 g_p ;  QR: -Q1^T r  (= -q1)
 *   tri[j]  upper 3x3 (6: 00 01 02 11 12 22) CHOLESKY  L^T (unit diag) ; QR: R1
 * so that  S_ab = delta_ab (lambda I + sum A^T A) - sum_j Z_a diag(dinv) Z_b^T
 *          rhs_a = g_c[a] - sum_{i in cam a} Z_i (dinv o t)
 *          dx_p = tri^-1 (dinv o (t - sum_i Z_i^T dx_c[cam_i])). */
typedef struct {
    S *Z, *dinv, *t, *tri;
    int *perm; /* 3 per point (may be NULL = identity): position c of the point's pivoted 3x3 block is coordinate perm[c] */
} FN(elim_t);

/* CHOLESKY: block elimination of the point variables from (J^T J + lambda I) -- identical to LDL^T of
 * the whole matrix (BacktrackLevMarqCholesky.h:274-282) with the point columns ordered first. */
static void FN(elim_cholesky)(int M, const int *pt_ptr, const S *Jc, const S *Jp, const S *fvec, S lambda,
                              FN(elim_t) * e)
{
    for (int j = 0; j < M; j++) {
        S U[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
        for (int i = pt_ptr[j]; i < pt_ptr[j + 1]; i++) {
            const S *B = Jp + 6 * (size_t)i;
            const S *r = fvec + 2 * (size_t)i;
            U[0] += B[0] * B[0] + B[3] * B[3];
            U[1] += B[0] * B[1] + B[3] * B[4];
            U[2] += B[0] * B[2] + B[3] * B[5];
            U[3] += B[1] * B[1] + B[4] * B[4];
            U[4] += B[1] * B[2] + B[4] * B[5];
            U[5] += B[2] * B[2] + B[5] * B[5];
            for (int c = 0; c < 3; c++) gp[c] -= B[c] * r[0] + B[3 + c] * r[1];
        }
        const S d0 = U[0] + lambda;
        const S l10 = U[1] / d0, l20 = U[2] / d0;
        const S d1 = (U[3] + lambda) - l10 * l10 * d0;
        const S l21 = (U[4] - l20 * l10 * d0) / d1;
        const S d2 = (U[5] + lambda) - l20 * l20 * d0 - l21 * l21 * d1;
        S *dinv = e->dinv + 3 * (size_t)j, *t = e->t + 3 * (size_t)j, *tri = e->tri + 6 * (size_t)j;
        dinv[0] = (S)1.0 / d0; dinv[1] = (S)1.0 / d1; dinv[2] = (S)1.0 / d2;
        t[0] = gp[0]; t[1] = gp[1] - l10 * t[0]; t[2] = gp[2] - l20 * t[0] - l21 * t[1];
        tri[0] = 1; tri[1] = l10; tri[2] = l20; tri[3] = 1; tri[4] = l21; tri[5] = 1;
        for (int i = pt_ptr[j]; i < pt_ptr[j + 1]; i++) {
            const S *A = Jc + 18 * (size_t)i, *B = Jp + 6 * (size_t)i;
            S Bt[6]; /* B L^-T, 2x3 */
            for (int r = 0; r < 2; r++) {
                Bt[3 * r] = B[3 * r];
                Bt[3 * r + 1] = B[3 * r + 1] - l10 * Bt[3 * r];
                Bt[3 * r + 2] = B[3 * r + 2] - l20 * Bt[3 * r] - l21 * Bt[3 * r + 1];
            }
            S *Z = e->Z + 27 * (size_t)i;
            for (int c = 0; c < 9; c++)
                for (int m = 0; m < 3; m++) Z[3 * c + m] = A[c] * Bt[m] + A[9 + c] * Bt[3 + m];
        }
    }
}

/* QRCHOL / QRKIT left block: per point, unpivoted Householder QR of [(Jp)_j ; sqrt(lambda) I3]
 * ((2k_j+3) x 3), the block BlockDiagonalSparseQR factors (BAFunctor.h:99-105, BAFunctor.cpp:64-68,
 * BacktrackLevMarqQRChol.h:291-319).  Deviation (documented in DESIGN.md): the reference's dense block
 * solver is ColPivHouseholderQR; with the sqrt(lambda) rows the block has full rank and the unpivoted
 * factorisation solves the same least-squares problem.
 * Q1 (thin Q, (2k+3) x 3) is formed explicitly; R12_i = Q1_i^T A_i ; q1 = Q1^T [r;0]. */
static void FN(elim_qr)(int M, const int *pt_ptr, const S *Jc, const S *Jp, const S *fvec, S lambda, FN(elim_t) * e,
                        S *Q1obs /* K x 6 (2x3 per obs) or NULL */, S *Q1lam /* M x 9 or NULL */)
{
    const S sl = SQRT(lambda);
    int kmax = 0;
    for (int j = 0; j < M; j++)
        if (pt_ptr[j + 1] - pt_ptr[j] > kmax) kmax = pt_ptr[j + 1] - pt_ptr[j];
    const int mmax = 2 * kmax + 3;
    S *Wk = (S *)malloc(sizeof(S) * 3 * (size_t)mmax); /* column-major m x 3 */
    S *Q = (S *)malloc(sizeof(S) * 3 * (size_t)mmax);
    for (int j = 0; j < M; j++) {
        const int i0 = pt_ptr[j], k = pt_ptr[j + 1] - i0, m = 2 * k + 3;
        for (int i = 0; i < k; i++) {
            const S *B = Jp + 6 * (size_t)(i0 + i);
            for (int c = 0; c < 3; c++) {
                Wk[c * m + 2 * i] = B[c];
                Wk[c * m + 2 * i + 1] = B[3 + c];
            }
        }
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 3; r++) Wk[c * m + 2 * k + r] = (r == c) ? sl : (S)0;
        S tau[3];
        int perm[3] = {0, 1, 2};
        for (int c = 0; c < 3; c++) {
            /* ColPivHouseholderQR (BAFunctor.h:99,104): the remaining column of largest norm (rows c..m-1) comes next; the
             * first maximum wins a tie; colsPermutation() is undone in backsub (BacktrackLevMarqQRChol.h:360) */
            {
                int best = c;
                S nbest = -1;
                for (int c2 = c; c2 < 3; c2++) {
                    S nn = 0;
                    for (int r = c; r < m; r++) nn += Wk[c2 * m + r] * Wk[c2 * m + r];
                    if (nn > nbest) { nbest = nn; best = c2; }
                }
                if (best != c) {
                    for (int r = 0; r < m; r++) { const S t_ = Wk[c * m + r]; Wk[c * m + r] = Wk[best * m + r]; Wk[best * m + r] = t_; }
                    const int p_ = perm[c]; perm[c] = perm[best]; perm[best] = p_;
                }
            }
            S *col = Wk + c * m;
            S xn = 0;
            for (int r = c + 1; r < m; r++) xn += col[r] * col[r];
            const S alpha = col[c];
            S beta = SQRT(alpha * alpha + xn);
            if (beta == (S)0) { /* zero column (only with lambda = 0, MOREQR stage 1): identity reflector */
                tau[c] = 0;
                continue;
            }
            if (alpha > 0) beta = -beta;
            tau[c] = (beta - alpha) / beta;
            const S sc = (S)1.0 / (alpha - beta);
            for (int r = c + 1; r < m; r++) col[r] *= sc;
            col[c] = beta;
            for (int c2 = c + 1; c2 < 3; c2++) {
                S *cc = Wk + c2 * m;
                S w = cc[c];
                for (int r = c + 1; r < m; r++) w += col[r] * cc[r];
                w *= tau[c];
                cc[c] -= w;
                for (int r = c + 1; r < m; r++) cc[r] -= col[r] * w;
            }
        }
        /* thin Q = H0 H1 H2 [I3;0] */
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < m; r++) Q[c * m + r] = (r == c) ? (S)1 : (S)0;
        for (int h = 2; h >= 0; h--) {
            const S *v = Wk + h * m;
            for (int c = 0; c < 3; c++) {
                S *qc = Q + c * m;
                S w = qc[h];
                for (int r = h + 1; r < m; r++) w += v[r] * qc[r];
                w *= tau[h];
                qc[h] -= w;
                for (int r = h + 1; r < m; r++) qc[r] -= v[r] * w;
            }
        }
        S *dinv = e->dinv + 3 * (size_t)j, *t = e->t + 3 * (size_t)j, *tri = e->tri + 6 * (size_t)j;
        dinv[0] = dinv[1] = dinv[2] = 1;
        tri[0] = Wk[0]; tri[1] = Wk[m]; tri[2] = Wk[2 * m]; tri[3] = Wk[m + 1]; tri[4] = Wk[2 * m + 1];
        tri[5] = Wk[2 * m + 2];
        if (e->perm) { e->perm[3 * (size_t)j] = perm[0]; e->perm[3 * (size_t)j + 1] = perm[1]; e->perm[3 * (size_t)j + 2] = perm[2]; }
        S q1[3] = {0, 0, 0};
        for (int i = 0; i < k; i++) {
            const S *A = Jc + 18 * (size_t)(i0 + i);
            const S *r = fvec + 2 * (size_t)(i0 + i);
            S Qi[6]; /* 2x3 row-major */
            for (int c = 0; c < 3; c++) {
                Qi[c] = Q[c * m + 2 * i];
                Qi[3 + c] = Q[c * m + 2 * i + 1];
                q1[c] += Qi[c] * r[0] + Qi[3 + c] * r[1];
            }
            if (Q1obs)
                for (int c = 0; c < 6; c++) Q1obs[6 * (size_t)(i0 + i) + c] = Qi[c];
            S *Z = e->Z + 27 * (size_t)(i0 + i);
            for (int c = 0; c < 9; c++)
                for (int mm = 0; mm < 3; mm++) Z[3 * c + mm] = A[c] * Qi[mm] + A[9 + c] * Qi[3 + mm];
        }
        if (Q1lam)
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 3; r++) Q1lam[9 * (size_t)j + 3 * r + c] = Q[c * m + 2 * k + r];
        for (int c = 0; c < 3; c++) t[c] = -q1[c];
    }
    free(Wk);
    free(Q);
}

/* MOREQR (src/Eigen_ext/BacktrackLevMarqMore.h:204-425, README.md:16 "performing 2 QR decompositions in each step"):
 * stage 1, once per outer iteration (:288 m_solver.compute(J)): per point the Householder QR of (Jp)_j (2k_j x 3) ->
 *   R1_j, thin Q1, R12_i = Q1_i^T A_i, q1 = Q1^T r.  Computed by elim_qr with lambda = 0, i.e. as the QR of
 *   [(Jp)_j ; 0_3]: the same factorisation, and no special cases for points with fewer than two observations (a
 *   column that is already zero gets the identity reflector and a zero diagonal entry of R1);
 * stage 2, per trial (:297-345 QR of [R ; sqrt(lambda) I]): per point the QR of [sqrt(lambda) I3 ; R1_j] (6 x 3) ->
 *   Rt1_j and the 3x3 block QR of its thin Q that multiplies the R1 rows; Z_i = R12_i^T QR, t = QR^T (-q1).
 * The reduced camera system is then the same expression as for QRCHOL (the algebra is in DESIGN.md). */
typedef struct {
    S *R12T; /* K x 27: R12_i^T (9x3 row-major) */
    S *R1;   /* M x 6 upper triangle 00 01 02 11 12 22 */
    S *mq1;  /* M x 3: -q1 */
    int *perm; /* M x 3: column permutation of the outer factorisation */
} FN(more_t);

static void FN(more_outer)(int M, const int *pt_ptr, const S *Jc, const S *Jp, const S *fvec, FN(more_t) * o, S *Q1obs, S *Q1lam)
{
    FN(elim_t) e;
    int K = pt_ptr[M];
    e.Z = o->R12T; e.tri = o->R1; e.t = o->mq1; e.perm = o->perm;
    e.dinv = (S *)malloc(sizeof(S) * 3 * (size_t)(M > 0 ? M : 1));
    (void)K;
    FN(elim_qr)(M, pt_ptr, Jc, Jp, fvec, (S)0, &e, Q1obs, Q1lam); /* lambda = 0: QR of [B;0] */
    free(e.dinv);
}

static void FN(more_trial)(int M, const int *pt_ptr, S lambda, const FN(more_t) * o, FN(elim_t) * e, S *mQl /* M x 9 or NULL */, S *mQR /* M x 9 or NULL */)
{
    const S sl = SQRT(lambda);
    for (int j = 0; j < M; j++) {
        /* QR of [sl I3 ; R1] (6x3), lambda rows first; column-major work array W[c][row] */
        const S *R1 = o->R1 + 6 * (size_t)j;
        S W[3][6] = {{sl, 0, 0, R1[0], 0, 0}, {0, sl, 0, R1[1], R1[3], 0}, {0, 0, sl, R1[2], R1[4], R1[5]}};
        S tau[3], Rt[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int c = 0; c < 3; c++) {
            S xn = 0;
            for (int r = 3; r < 6; r++) xn += W[c][r] * W[c][r];
            const S alpha = sl; /* the lambda row c is untouched by the earlier reflectors */
            const S beta = -SQRT(alpha * alpha + xn);
            tau[c] = (beta - alpha) / beta;
            const S sc = (S)1.0 / (alpha - beta);
            for (int r = 3; r < 6; r++) W[c][r] *= sc;
            Rt[c][c] = beta;
            for (int c2 = c + 1; c2 < 3; c2++) {
                S w = 0;
                for (int r = 3; r < 6; r++) w += W[c][r] * W[c2][r];
                w *= tau[c];
                Rt[c][c2] = -w;
                for (int r = 3; r < 6; r++) W[c2][r] -= W[c][r] * w;
            }
        }
        /* thin Q (6 x 3): lambda rows Ql[h][c], R1 rows QR[r][c] */
        S Ql[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, QR[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int h = 2; h >= 0; h--)
            for (int c = 0; c < 3; c++) {
                S w = Ql[h][c];
                for (int r = 0; r < 3; r++) w += W[h][3 + r] * QR[r][c];
                w *= tau[h];
                Ql[h][c] -= w;
                for (int r = 0; r < 3; r++) QR[r][c] -= W[h][3 + r] * w;
            }
        if (mQl)
            for (int h = 0; h < 3; h++)
                for (int c = 0; c < 3; c++) { mQl[9 * (size_t)j + 3 * h + c] = Ql[h][c]; mQR[9 * (size_t)j + 3 * h + c] = QR[h][c]; }
        S *dinv = e->dinv + 3 * (size_t)j, *t = e->t + 3 * (size_t)j, *tri = e->tri + 6 * (size_t)j;
        const S *mq1 = o->mq1 + 3 * (size_t)j;
        dinv[0] = dinv[1] = dinv[2] = 1;
        for (int c = 0; c < 3; c++) t[c] = QR[0][c] * mq1[0] + QR[1][c] * mq1[1] + QR[2][c] * mq1[2];
        tri[0] = Rt[0][0]; tri[1] = Rt[0][1]; tri[2] = Rt[0][2]; tri[3] = Rt[1][1]; tri[4] = Rt[1][2]; tri[5] = Rt[2][2];
        for (int i = pt_ptr[j]; i < pt_ptr[j + 1]; i++) {
            const S *Z0 = o->R12T + 27 * (size_t)i;
            S *Z = e->Z + 27 * (size_t)i;
            for (int c = 0; c < 9; c++)
                for (int m = 0; m < 3; m++) Z[3 * c + m] = Z0[3 * c] * QR[0][m] + Z0[3 * c + 1] * QR[1][m] + Z0[3 * c + 2] * QR[2][m];
        }
    }
}

/* Reduced camera system from the elimination workspace.
 * Smat: D x D column-major (full symmetric), rhs: D, gc: D (camera part of g = -J^T r).
 * QRCHOL: S = J2bot^T J2bot, rhs = -J2bot^T qtb2 (BacktrackLevMarqQRChol.h:334-341), evaluated as
 * (A^T A + lambda I) - R12^T R12 and g_c + R12^T q1 -- algebraically identical because Q is orthogonal.
 * CHOLESKY: the Schur complement of the point block of J^T J + lambda I. */
#ifndef ORA_WIDE_SUMS_DECLARED
#define ORA_WIDE_SUMS_DECLARED
/* Experiment switch (default 0 = the oracle as committed): the long sums of the reduced camera system -- S, its rhs and g_c, thousands
 * of terms per entry, which this restatement (like a sparse product) adds one after the other -- are accumulated in long double and
 * rounded once.  Nothing else changes.  Used by tests/golden/make_referee.py (ensemble_*_widesums) to show which part of the fp64
 * oracle's rounding noise decides where its free run stops. */
static int ora_wide_sums_flag = 0;
void ora_set_wide_sums(int on) { ora_wide_sums_flag = on; }
/* MOREQR's right block QR only (solve_more_qr) instead of the LDL^T of S = (Jc'Jc + lambda I) - sum Z Z': off by default, like the
 * product's BA_MOREQR_QR switch (round 4: the route exists on both sides and agrees to 1e-7 / 1e-10 on the first step; the product's
 * dense QR kernels have a sporadic accuracy defect that keeps it from being the default). */
static int ora_more_qr_flag = 1; /* MOREQR: QR-only right block (the reference's route); 0 = LDL^T of the same reduced system */
void ora_set_more_qr(int on) { ora_more_qr_flag = on; }
#endif
static void FN(build_reduced)(int N, int K, const int *cam_idx, const int *pt_idx, const int *pt_ptr, int M,
                              const S *Jc, const S *fvec, S lambda, const FN(elim_t) * e, S *Smat, S *rhs, S *gc)
{
    const int D = 9 * N;
    if (ora_wide_sums_flag) {
        long double *W = (long double *)calloc((size_t)D * D + 2 * (size_t)D, sizeof(long double));
        long double *wr = W + (size_t)D * D, *wg = wr + D;
        for (int i = 0; i < K; i++) {
            const S *A = Jc + 18 * (size_t)i;
            const S *r = fvec + 2 * (size_t)i;
            const int a = cam_idx[i];
            for (int c = 0; c < 9; c++) wg[9 * a + c] -= A[c] * r[0] + A[9 + c] * r[1];
            for (int c = 0; c < 9; c++)
                for (int c2 = 0; c2 < 9; c2++) W[(size_t)(9 * a + c2) * D + 9 * a + c] += A[c] * A[c2] + A[9 + c] * A[9 + c2];
        }
        for (int j = 0; j < M; j++) {
            const S *dinv = e->dinv + 3 * (size_t)j, *t = e->t + 3 * (size_t)j;
            const S td[3] = {dinv[0] * t[0], dinv[1] * t[1], dinv[2] * t[2]};
            for (int ia = pt_ptr[j]; ia < pt_ptr[j + 1]; ia++) {
                const S *Za = e->Z + 27 * (size_t)ia;
                const int a = cam_idx[ia];
                for (int c = 0; c < 9; c++) wr[9 * a + c] -= Za[3 * c] * td[0] + Za[3 * c + 1] * td[1] + Za[3 * c + 2] * td[2];
                for (int ib = pt_ptr[j]; ib < pt_ptr[j + 1]; ib++) {
                    const S *Zb = e->Z + 27 * (size_t)ib;
                    const int b = cam_idx[ib];
                    for (int c = 0; c < 9; c++) {
                        const S z0 = Za[3 * c] * dinv[0], z1 = Za[3 * c + 1] * dinv[1], z2 = Za[3 * c + 2] * dinv[2];
                        for (int c2 = 0; c2 < 9; c2++)
                            W[(size_t)(9 * b + c2) * D + 9 * a + c] -= z0 * Zb[3 * c2] + z1 * Zb[3 * c2 + 1] + z2 * Zb[3 * c2 + 2];
                    }
                }
            }
        }
        for (int c = 0; c < D; c++) {
            W[(size_t)c * D + c] += lambda;
            wr[c] += wg[c];
        }
        for (size_t q = 0; q < (size_t)D * D; q++) Smat[q] = (S)W[q];
        for (int c = 0; c < D; c++) { rhs[c] = (S)wr[c]; gc[c] = (S)wg[c]; }
        free(W);
        (void)pt_idx;
        return;
    }

    memset(Smat, 0, sizeof(S) * (size_t)D * D);
    for (int c = 0; c < D; c++) { rhs[c] = 0; gc[c] = 0; }
    for (int i = 0; i < K; i++) {
        const S *A = Jc + 18 * (size_t)i;
        const S *r = fvec + 2 * (size_t)i;
        const int a = cam_idx[i];
        for (int c = 0; c < 9; c++) gc[9 * a + c] -= A[c] * r[0] + A[9 + c] * r[1];
        for (int c = 0; c < 9; c++)
            for (int c2 = 0; c2 < 9; c2++)
                Smat[(size_t)(9 * a + c2) * D + 9 * a + c] += A[c] * A[c2] + A[9 + c] * A[9 + c2];
    }
    for (int j = 0; j < M; j++) {
        const S *dinv = e->dinv + 3 * (size_t)j, *t = e->t + 3 * (size_t)j;
        const S td[3] = {dinv[0] * t[0], dinv[1] * t[1], dinv[2] * t[2]};
        for (int ia = pt_ptr[j]; ia < pt_ptr[j + 1]; ia++) {
            const S *Za = e->Z + 27 * (size_t)ia;
            const int a = cam_idx[ia];
            for (int c = 0; c < 9; c++) rhs[9 * a + c] -= Za[3 * c] * td[0] + Za[3 * c + 1] * td[1] + Za[3 * c + 2] * td[2];
            for (int ib = pt_ptr[j]; ib < pt_ptr[j + 1]; ib++) {
                const S *Zb = e->Z + 27 * (size_t)ib;
                const int b = cam_idx[ib];
                for (int c = 0; c < 9; c++) {
                    const S z0 = Za[3 * c] * dinv[0], z1 = Za[3 * c + 1] * dinv[1], z2 = Za[3 * c + 2] * dinv[2];
                    for (int c2 = 0; c2 < 9; c2++)
                        Smat[(size_t)(9 * b + c2) * D + 9 * a + c] -= z0 * Zb[3 * c2] + z1 * Zb[3 * c2 + 1] + z2 * Zb[3 * c2 + 2];
                }
            }
        }
    }
    (void)pt_idx;
    for (int c = 0; c < D; c++) {
        Smat[(size_t)c * D + c] += lambda;
        rhs[c] += gc[c];
    }
}

/* Back-substitution for the point steps (BacktrackLevMarqQRChol.h:343-360). dx = [3M points | 9N cameras]. */
static void FN(backsub)(int M, const int *pt_ptr, const int *cam_idx, const FN(elim_t) * e, S *dx)
{
    const S *dxc = dx + 3 * (size_t)M;
    for (int j = 0; j < M; j++) {
        const S *dinv = e->dinv + 3 * (size_t)j, *t = e->t + 3 * (size_t)j, *tri = e->tri + 6 * (size_t)j;
        S u[3] = {t[0], t[1], t[2]};
        for (int i = pt_ptr[j]; i < pt_ptr[j + 1]; i++) {
            const S *Z = e->Z + 27 * (size_t)i;
            const S *dc = dxc + 9 * (size_t)cam_idx[i];
            for (int c = 0; c < 9; c++) {
                u[0] -= Z[3 * c] * dc[c];
                u[1] -= Z[3 * c + 1] * dc[c];
                u[2] -= Z[3 * c + 2] * dc[c];
            }
        }
        u[0] *= dinv[0]; u[1] *= dinv[1]; u[2] *= dinv[2];
        const S x2 = u[2] / tri[5];
        const S x1 = (u[1] - tri[4] * x2) / tri[3];
        const S x0 = (u[0] - tri[1] * x1 - tri[2] * x2) / tri[0];
        if (e->perm) { /* m_dx = colsPermutation() * m_dx, BacktrackLevMarqQRChol.h:360 */
            const int *pp = e->perm + 3 * (size_t)j;
            dx[3 * (size_t)j + pp[0]] = x0; dx[3 * (size_t)j + pp[1]] = x1; dx[3 * (size_t)j + pp[2]] = x2;
        } else {
            dx[3 * (size_t)j] = x0; dx[3 * (size_t)j + 1] = x1; dx[3 * (size_t)j + 2] = x2;
        }
    }
}

/* QRKIT right block: dense thin QR of J2bot (BAFunctor.h:101, README.md:14).
 * J2bot = rows 3.. of every point block of Q^T [Jc;0], followed by sqrt(lambda) I_D; the rhs is qtb2.
 * Built explicitly: for point j with thin Q1 (rows Q1_i per observation, Q1lam for the lambda rows) the
 * projected rows are  P_j [A;0] with P_j = Q2^T; since only the span matters for the least-squares solve,
 * the oracle uses the equivalent (I - Q1 Q1^T)[A;0] rows (same R up to an orthogonal row transform).
 * Sized for small problems only (dense (2K+3M+D) x D). */
/* J2bot (m = 2K + 3M + D rows) and b2 = -qtb2 from the point blocks' thin Q: shared by QRKIT (lambda > 0) and by MOREQR's outer
 * factorisation (lambda = 0: the camera rows are zero). */
static void FN(build_j2bot)(int N, int M, int K, const int *cam_idx, const int *pt_ptr, const S *Jc, const S *fvec,
                            S lambda, const S *Q1obs, const S *Q1lam, S *A2, S *b2)
{
    const int D = 9 * N;
    const size_t m = 2 * (size_t)K + 3 * (size_t)M + D;
    for (int j = 0; j < M; j++) {
        const int i0 = pt_ptr[j], k = pt_ptr[j + 1] - i0;
        const size_t row0 = 2 * (size_t)i0 + 3 * (size_t)j;
        /* q1 = Q1^T [r;0] ; per camera column block: R12_a = Q1_a^T A_a */
        S q1[3] = {0, 0, 0};
        for (int i = 0; i < k; i++) {
            const S *Qi = Q1obs + 6 * (size_t)(i0 + i);
            const S *r = fvec + 2 * (size_t)(i0 + i);
            for (int c = 0; c < 3; c++) q1[c] += Qi[c] * r[0] + Qi[3 + c] * r[1];
        }
        for (int ia = 0; ia < k; ia++) {
            const S *A = Jc + 18 * (size_t)(i0 + ia);
            const S *Qa = Q1obs + 6 * (size_t)(i0 + ia);
            const int a = cam_idx[i0 + ia];
            S R12[27]; /* 3x9 */
            for (int mm = 0; mm < 3; mm++)
                for (int c = 0; c < 9; c++) R12[9 * mm + c] = Qa[mm] * A[c] + Qa[3 + mm] * A[9 + c];
            for (int c = 0; c < 9; c++) {
                S *col = A2 + (size_t)(9 * a + c) * m + row0;
                for (int ib = 0; ib < k; ib++) {
                    const S *Qb = Q1obs + 6 * (size_t)(i0 + ib);
                    for (int rr = 0; rr < 2; rr++) {
                        S v = (ib == ia) ? A[9 * rr + c] : (S)0;
                        v -= Qb[3 * rr] * R12[c] + Qb[3 * rr + 1] * R12[9 + c] + Qb[3 * rr + 2] * R12[18 + c];
                        col[2 * ib + rr] += v;
                    }
                }
                const S *Ql = Q1lam + 9 * (size_t)j;
                for (int rr = 0; rr < 3; rr++)
                    col[2 * k + rr] -= Ql[3 * rr] * R12[c] + Ql[3 * rr + 1] * R12[9 + c] + Ql[3 * rr + 2] * R12[18 + c];
            }
        }
        for (int ib = 0; ib < k; ib++) {
            const S *Qb = Q1obs + 6 * (size_t)(i0 + ib);
            const S *r = fvec + 2 * (size_t)(i0 + ib);
            for (int rr = 0; rr < 2; rr++)
                b2[row0 + 2 * ib + rr] = r[rr] - (Qb[3 * rr] * q1[0] + Qb[3 * rr + 1] * q1[1] + Qb[3 * rr + 2] * q1[2]);
        }
        const S *Ql = Q1lam + 9 * (size_t)j;
        for (int rr = 0; rr < 3; rr++)
            b2[row0 + 2 * k + rr] = -(Ql[3 * rr] * q1[0] + Ql[3 * rr + 1] * q1[1] + Ql[3 * rr + 2] * q1[2]);
    }
    const S sl = SQRT(lambda);
    for (int c = 0; c < D; c++) A2[(size_t)c * m + 2 * (size_t)K + 3 * (size_t)M + c] = sl;
    /* min || J2bot dx_c + qtb2 || */
    for (size_t r = 0; r < m; r++) b2[r] = -b2[r];
}

static int FN(solve_reduced_qr)(int N, int M, int K, const int *cam_idx, const int *pt_ptr, const S *Jc, const S *fvec,
                                S lambda, const S *Q1obs, const S *Q1lam, S *dxc)
{
    const int D = 9 * N;
    const size_t m = 2 * (size_t)K + 3 * (size_t)M + D;
    S *A2 = (S *)calloc(m * D, sizeof(S));
    S *b2 = (S *)calloc(m, sizeof(S));
    if (!A2 || !b2) { free(A2); free(b2); return -1; }
    FN(build_j2bot)(N, M, K, cam_idx, pt_ptr, Jc, fvec, lambda, Q1obs, Q1lam, A2, b2);
    FN(dense_qr_solve)((int)m, D, A2, b2, dxc);
    free(A2);
    free(b2);
    return 0;
}

/* MOREQR's right block, QR only (BacktrackLevMarqMore.h:288-345; round 4 -- rounds 1 - 3 formed S = (Jc'Jc + lambda I) - sum Z Z' and
 * factored it by LDL^T: the same camera step in exact arithmetic, the conditioning of the normal equations in floating point).
 *   outer (:288-291, m_solver.compute(J), once per outer iteration; recomputed per call here): the block-angular QR of J -- the point
 *     blocks by more_outer (lambda = 0), then the dense QR of J2bot(lambda = 0) -> R22 (D x D), c2 = the head of Q^T (-qtb2);
 *   inner (:297-345, m_solverInner.compute([R ; sqrt(lambda) I]) per trial): the point blocks [sqrt(lambda) I3 ; R1_j] by more_trial
 *     (thin Q = [Ql ; QR], 6 x 3), then the dense QR of the rows left for the camera columns:
 *       per point the complement (I - Q Q^T) [0 ; R12_j] (6 rows; rhs (I - Q Q^T) [0 ; -q1_j]) -- the rows orthogonal to the point's
 *       thin Q up to an orthogonal row transform, like J2bot --, R22 (rhs c2) and sqrt(lambda) I_D (rhs 0);
 *     dx_c from R y = Q^T rhs (the rhs columns carry the NEGATIVE residual parts throughout), the points by backsub.
 * No normal equations anywhere. */
static int FN(solve_more_qr)(int N, int M, int K, const int *cam_idx, const int *pt_ptr, const S *Jc, const S *fvec, S lambda,
                             const FN(more_t) * o, const S *Q1obs, const S *Q1lam, const FN(elim_t) * e, const S *mQl, const S *mQR, S *dxc)
{
    const int D = 9 * N;
    int rc = -1;
    const size_t mo = 2 * (size_t)K + 3 * (size_t)M + D, mi = 6 * (size_t)M + 2 * (size_t)D;
    S *R22 = (S *)malloc(sizeof(S) * (size_t)D * D), *c2 = (S *)malloc(sizeof(S) * (size_t)D);
    S *A2 = (S *)calloc(mo * D, sizeof(S)), *b2 = (S *)calloc(mo, sizeof(S));
    S *Ai = NULL, *bi = NULL, *Ri = NULL, *ci = NULL;
    if (!R22 || !c2 || !A2 || !b2) goto done;
    FN(build_j2bot)(N, M, K, cam_idx, pt_ptr, Jc, fvec, (S)0, Q1obs, Q1lam, A2, b2);
    if (FN(dense_qr_factor_tsqr)(mo, D, A2, b2, R22, c2)) goto done;
    free(A2); free(b2); A2 = b2 = NULL;
    Ai = (S *)calloc(mi * D, sizeof(S)); bi = (S *)calloc(mi, sizeof(S));
    Ri = (S *)malloc(sizeof(S) * (size_t)D * D); ci = (S *)malloc(sizeof(S) * (size_t)D);
    if (!Ai || !bi || !Ri || !ci) goto done;
    for (int j = 0; j < M; j++) {
        const S *Ql = mQl + 9 * (size_t)j, *QR = mQR + 9 * (size_t)j;
        const size_t row0 = 6 * (size_t)j;
        for (int i = pt_ptr[j]; i < pt_ptr[j + 1]; i++) {
            const S *Z0 = o->R12T + 27 * (size_t)i, *Z = e->Z + 27 * (size_t)i; /* 9 x 3: Z0[c][r] = R12_j(r, c); Z = Z0 QR */
            const int a = cam_idx[i];
            for (int c = 0; c < 9; c++) {
                S *col = Ai + (size_t)(9 * a + c) * mi + row0;
                for (int rr = 0; rr < 3; rr++) {
                    col[rr] -= Ql[3 * rr] * Z[3 * c] + Ql[3 * rr + 1] * Z[3 * c + 1] + Ql[3 * rr + 2] * Z[3 * c + 2];
                    col[3 + rr] += Z0[3 * c + rr] - (QR[3 * rr] * Z[3 * c] + QR[3 * rr + 1] * Z[3 * c + 1] + QR[3 * rr + 2] * Z[3 * c + 2]);
                }
            }
        }
        const S *mq1 = o->mq1 + 3 * (size_t)j, *t = e->t + 3 * (size_t)j; /* t = QR^T (-q1) */
        for (int rr = 0; rr < 3; rr++) {
            bi[row0 + rr] = -(Ql[3 * rr] * t[0] + Ql[3 * rr + 1] * t[1] + Ql[3 * rr + 2] * t[2]);
            bi[row0 + 3 + rr] = mq1[rr] - (QR[3 * rr] * t[0] + QR[3 * rr + 1] * t[1] + QR[3 * rr + 2] * t[2]);
        }
    }
    {
        const S sl = SQRT(lambda);
        const size_t rR = 6 * (size_t)M, rL = rR + (size_t)D;
        for (int c = 0; c < D; c++) {
            for (int i = 0; i <= c; i++) Ai[(size_t)c * mi + rR + i] = R22[(size_t)c * D + i];
            bi[rR + c] = c2[c];
            Ai[(size_t)c * mi + rL + c] = sl;
        }
    }
    if (FN(dense_qr_factor_tsqr)(mi, D, Ai, bi, Ri, ci)) goto done;
    for (int j = D - 1; j >= 0; j--) {
        S a = ci[j];
        for (int c = j + 1; c < D; c++) a -= Ri[(size_t)c * D + j] * dxc[c];
        dxc[j] = a / Ri[(size_t)j * D + j];
    }
    rc = 0;
done:
    free(R22); free(c2); free(A2); free(b2); free(Ai); free(bi); free(Ri); free(ci);
    return rc;
}

/* QRSPQR (kind 4): SuiteSparseQR on the WHOLE [J ; sqrt(lambda) I] (typedef SPQR<JacobianType> SchurlikeQRSolver, BAFunctor.h:113-116;
 * lm.minimize with that solver, bundle_adjustment_large.cpp:151-157; README.md:17 "QR decomposition on full Jacobian").  SuiteSparse
 * is absent, so what is restated is the definition: a Householder QR of the whole (2K + 3M + 9N) x (3M + 9N) matrix -- all point
 * AND camera columns, natural order, no block elimination, no fill-reducing ordering (an ordering changes R, not the least-squares
 * solution) -- and  dx = argmin || [J ; sqrt(lambda) I] dx + [r ; 0] ||  from R dx = -Q^T [r ; 0].  Dense storage: small problems only. */
static int FN(solve_whole_qr)(int N, int M, int K, const int *cam_idx, const int *pt_idx, const S *Jc, const S *Jp, const S *fvec,
                              S lambda, S *dx)
{
    const size_t n = 3 * (size_t)M + 9 * (size_t)N, m = 2 * (size_t)K + n;
    S *A = (S *)calloc(m * n, sizeof(S));
    S *b = (S *)calloc(m, sizeof(S));
    if (!A || !b) { free(A); free(b); return -1; }
    for (int i = 0; i < K; i++) { /* rows 2i, 2i+1: the 2 x 3 point block and the 2 x 9 camera block of observation i (BAFunctor.h:265-292) */
        const S *Ac = Jc + 18 * (size_t)i, *Bp = Jp + 6 * (size_t)i;
        for (int rr = 0; rr < 2; rr++) {
            for (int c = 0; c < 3; c++) A[(3 * (size_t)pt_idx[i] + c) * m + 2 * (size_t)i + rr] = Bp[3 * rr + c];
            for (int c = 0; c < 9; c++) A[(3 * (size_t)M + 9 * (size_t)cam_idx[i] + c) * m + 2 * (size_t)i + rr] = Ac[9 * rr + c];
            b[2 * (size_t)i + rr] = -fvec[2 * (size_t)i + rr];
        }
    }
    const S sl = SQRT(lambda);
    for (size_t c = 0; c < n; c++) A[c * m + 2 * (size_t)K + c] = sl;
    FN(dense_qr_solve)((int)m, (int)n, A, b, dx);
    free(A);
    free(b);
    return 0;
}

/* g = -J^T r (the reference's JtRes, BacktrackLevMarqQRChol.h:267 / ...Cholesky.h:250) and
 * max diag(J^T J) (squared column norms, ...QRChol.h:270-280; JtJ.diagonal().maxCoeff(), ...Cholesky.h:263-265). */
static void FN(grad_diag)(int N, int M, int K, const int *cam_idx, const int *pt_idx, const S *Jc, const S *Jp,
                          const S *fvec, S *gout, S *diagmax)
{
    const size_t np = 3 * (size_t)M + 9 * (size_t)N;
    S *dg = (S *)calloc(np, sizeof(S));
    if (gout)
        for (size_t c = 0; c < np; c++) gout[c] = 0;
    for (int i = 0; i < K; i++) {
        const S *A = Jc + 18 * (size_t)i, *B = Jp + 6 * (size_t)i, *r = fvec + 2 * (size_t)i;
        for (int c = 0; c < 3; c++) {
            if (gout) gout[3 * (size_t)pt_idx[i] + c] -= B[c] * r[0] + B[3 + c] * r[1];
            dg[3 * (size_t)pt_idx[i] + c] += B[c] * B[c] + B[3 + c] * B[3 + c];
        }
        for (int c = 0; c < 9; c++) {
            if (gout) gout[3 * (size_t)M + 9 * (size_t)cam_idx[i] + c] -= A[c] * r[0] + A[9 + c] * r[1];
            dg[3 * (size_t)M + 9 * (size_t)cam_idx[i] + c] += A[c] * A[c] + A[9 + c] * A[9 + c];
        }
    }
    S dm = 0;
    for (size_t c = 0; c < np; c++)
        if (dg[c] > dm) dm = dg[c];
    free(dg);
    if (diagmax) *diagmax = dm;
}

/* One LM trial's linear solve: dx (3M+9N) from J, r, lambda.  kind: 0 QRKIT, 1 QRCHOL, 2 CHOLESKY, 3 MOREQR, 4 QRSPQR.
 * Optional outputs (may be NULL): Sout D*D col-major, rhsout D, gout 3M+9N (= -J^T r, the reference's JtRes,
 * BacktrackLevMarqQRChol.h:267), diagmax = max diag(J^T J) (:270-280). */
int FN(ora_step)(int kind, int N, int M, int K, const int *cam_idx, const int *pt_idx, const S *Jc, const S *Jp,
                 const S *fvec, S lambda, S *dx, S *Sout, S *rhsout, S *gout, S *diagmax)
{
    const int D = 9 * N;
    /* kind | 256: assemble only (elimination + reduced system into Sout / rhsout), no factorisation, dx untouched --
     * for parity checks of the assembly at sizes where the plain-C dense LDL^T would take minutes (D = 9216) */
    const int assemble_only = (kind & 256) != 0;
    kind &= 255;
    int *pt_ptr = (int *)malloc(sizeof(int) * ((size_t)M + 1));
    /* observations must be sorted by point (BAL files are; BacktrackLevMarqQRChol.h:291-309 relies on it) */
    {
        int p = 0;
        pt_ptr[0] = 0;
        for (int i = 0; i < K; i++) {
            if (pt_idx[i] < p) { free(pt_ptr); return -2; }
            while (p < pt_idx[i]) pt_ptr[++p] = i;
        }
        while (p < M) pt_ptr[++p] = K;
    }
    FN(elim_t) e;
    e.Z = (S *)malloc(sizeof(S) * 27 * (size_t)K);
    e.dinv = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    e.t = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    e.tri = (S *)malloc(sizeof(S) * 6 * (size_t)M);
    e.perm = (int *)malloc(sizeof(int) * 3 * (size_t)(M > 0 ? M : 1));
    for (size_t q_ = 0; q_ < 3 * (size_t)M; q_++) e.perm[q_] = (int)(q_ % 3); /* identity unless a QR symbol pivots */
    S *Smat = (S *)malloc(sizeof(S) * (size_t)D * D);
    S *rhs = (S *)malloc(sizeof(S) * (size_t)D);
    S *gc = (S *)malloc(sizeof(S) * (size_t)D);
    S *Q1obs = NULL, *Q1lam = NULL, *mQl = NULL, *mQR = NULL;
    FN(more_t) mo;
    mo.R12T = NULL; mo.R1 = NULL; mo.mq1 = NULL; mo.perm = NULL;
    int rc = 0;
    if (kind == 2) {
        FN(elim_cholesky)(M, pt_ptr, Jc, Jp, fvec, lambda, &e);
    } else if (kind == 3) {
        /* the outer factorisation is redone per call here (the reference does it once per outer iteration,
         * BacktrackLevMarqMore.h:288; same numbers) */
        mo.R12T = (S *)malloc(sizeof(S) * 27 * (size_t)(K > 0 ? K : 1));
        mo.R1 = (S *)malloc(sizeof(S) * 6 * (size_t)(M > 0 ? M : 1));
        mo.mq1 = (S *)malloc(sizeof(S) * 3 * (size_t)(M > 0 ? M : 1));
        mo.perm = e.perm; /* the outer QR's column permutation carries through the inner QR to the step */
        Q1obs = (S *)malloc(sizeof(S) * 6 * (size_t)(K > 0 ? K : 1));
        Q1lam = (S *)malloc(sizeof(S) * 9 * (size_t)(M > 0 ? M : 1));
        mQl = (S *)malloc(sizeof(S) * 9 * (size_t)(M > 0 ? M : 1));
        mQR = (S *)malloc(sizeof(S) * 9 * (size_t)(M > 0 ? M : 1));
        FN(more_outer)(M, pt_ptr, Jc, Jp, fvec, &mo, Q1obs, Q1lam);
        FN(more_trial)(M, pt_ptr, lambda, &mo, &e, mQl, mQR);
    } else {
        if (kind == 0) {
            Q1obs = (S *)malloc(sizeof(S) * 6 * (size_t)K);
            Q1lam = (S *)malloc(sizeof(S) * 9 * (size_t)M);
        }
        FN(elim_qr)(M, pt_ptr, Jc, Jp, fvec, lambda, &e, Q1obs, Q1lam);
    }
    FN(build_reduced)(N, K, cam_idx, pt_idx, pt_ptr, M, Jc, fvec, lambda, &e, Smat, rhs, gc);
    if (Sout) memcpy(Sout, Smat, sizeof(S) * (size_t)D * D);
    if (rhsout) memcpy(rhsout, rhs, sizeof(S) * (size_t)D);
    S *dxc = dx + 3 * (size_t)M;
    if (assemble_only) {
        /* nothing to solve */
    } else if (kind == 4) { /* the whole step from the whole-matrix QR (the reduced system above only serves Sout / rhsout) */
        rc = FN(solve_whole_qr)(N, M, K, cam_idx, pt_idx, Jc, Jp, fvec, lambda, dx);
    } else if (kind == 0) {
        rc = FN(solve_reduced_qr)(N, M, K, cam_idx, pt_ptr, Jc, fvec, lambda, Q1obs, Q1lam, dxc);
    } else if (kind == 3 && ora_more_qr_flag) { /* QR only: the reduced system above only serves Sout / rhsout (what the default LDL^T form factors) */
        rc = FN(solve_more_qr)(N, M, K, cam_idx, pt_ptr, Jc, fvec, lambda, &mo, Q1obs, Q1lam, &e, mQl, mQR, dxc);
    } else {
        FN(dense_ldlt)(D, Smat);
        FN(dense_ldlt_solve)(D, Smat, rhs);
        for (int c = 0; c < D; c++) dxc[c] = rhs[c];
    }
    if (!assemble_only && kind != 4) FN(backsub)(M, pt_ptr, cam_idx, &e, dx);
    if (gout || diagmax) FN(grad_diag)(N, M, K, cam_idx, pt_idx, Jc, Jp, fvec, gout, diagmax);
    free(Q1obs); free(Q1lam); free(mQl); free(mQR);
    free(mo.R12T); free(mo.R1); free(mo.mq1);
    free(e.Z); free(e.dinv); free(e.t); free(e.tri); free(e.perm);
    free(Smat); free(rhs); free(gc); free(pt_ptr);
    return rc;
}

/* ---- LM outer loop -------------------------------------------------------------------- */

/* BacktrackLevMarqQRCHol::minimize (BacktrackLevMarqQRChol.h:204-436) and
 * BacktrackLevMarqCholesky::minimize (BacktrackLevMarqCholesky.h:190-361): same skeleton, different
 * inner solve.  QRKIT's loop (Eigen::BacktrackLevMarq) is not vendored; it reuses this skeleton.
 * lm = {lambda_min, lambda_max, increase_base, tol_fun}; max_iter / max_fun_ev as LMParams.
 * snap (optional): the state x of every trial, for tests that inject it into another implementation.
 * trace: max_trials rows of 8 doubles {iter, accepted, f, rho, lambda_printed, lambda_used, e_test, |dx|}
 * (one row per printed table line, BacktrackLevMarqQRChol.h:383,397).  Stops early (status Running = -1)
 * after max_trials rows.  cam15/pts are updated in place with the reference's quirk that the flat-line
 * exit happens BEFORE x = xTest (:419-428).  Returns the Status integer (:39-46). */
int FN(ora_minimize)(int kind, int N, int M, int K, const int *cam_idx, const int *pt_idx, const S *meas, S tau,
                     S *cam15, S *pts, const double *lm, int max_iter, int max_fun_ev, int max_trials, double *trace,
                     int *ntrials_out, double *snap /* NULL, or max_trials x (15N + 3M): the state x each trial starts from */)
{
    const S lam_min = (S)lm[0], lam_max = (S)lm[1], inc_base = (S)lm[2], tol_fun = (S)lm[3];
    const size_t np = 3 * (size_t)M + 9 * (size_t)N;
    S *fvec = (S *)malloc(sizeof(S) * 2 * (size_t)K);
    S *Jc = (S *)malloc(sizeof(S) * 18 * (size_t)K);
    S *Jp = (S *)malloc(sizeof(S) * 6 * (size_t)K);
    S *dx = (S *)malloc(sizeof(S) * np);
    S *g = (S *)malloc(sizeof(S) * np);
    S *camT = (S *)malloc(sizeof(S) * 15 * (size_t)N);
    S *ptsT = (S *)malloc(sizeof(S) * 3 * (size_t)M);
    S lambda = (S)1e-3, lambda_inc = inc_base;
    S hist[2] = {0, 0};
    int fun_evals = 0, iter = 0, status = -1, ntr = 0, stop = 0;
    S energy = 0;
    while (1) {
        iter++;
        if (iter > max_iter) { status = 3; break; }
        if (fun_evals > max_fun_ev) { status = 2; break; }
        energy = FN(ora_residuals)(N, M, K, cam15, pts, cam_idx, pt_idx, meas, tau, fvec);
        fun_evals++;
        FN(ora_jacobian)(N, M, K, cam15, pts, cam_idx, pt_idx, meas, tau, Jc, Jp);
        {
            S dmax = 0;
            FN(grad_diag)(N, M, K, cam_idx, pt_idx, Jc, Jp, fvec, g, &dmax);
            /* lambda0 = 1e-12 * max diag(J^T J) (BacktrackLevMarqQRChol.h:278-280; ...Cholesky.h:263-265);
             * MOREQR: 1e-6 * max column norm (BacktrackLevMarqMore.h:272-284) */
            if (iter == 1) lambda = kind == 3 ? (S)(1e-6 * sqrt((double)dmax)) : (S)(1e-12 * (double)dmax);
        }
        while (1) {
            if (ntr >= max_trials) { stop = 1; status = -1; break; }
            if (snap) {
                double *sn = snap + (size_t)ntr * (15 * (size_t)N + 3 * (size_t)M);
                for (size_t c = 0; c < 15 * (size_t)N; c++) sn[c] = (double)cam15[c];
                for (size_t c = 0; c < 3 * (size_t)M; c++) sn[15 * (size_t)N + c] = (double)pts[c];
            }
            const int rc = FN(ora_step)(kind, N, M, K, cam_idx, pt_idx, Jc, Jp, fvec, lambda, dx, NULL, NULL, NULL, NULL);
            if (rc) { status = -3; stop = 1; break; }
            FN(ora_retract)(N, M, cam15, pts, dx, camT, ptsT);
            const S e_test = FN(ora_residuals)(N, M, K, camT, ptsT, cam_idx, pt_idx, meas, tau, NULL);
            fun_evals++;
            S dxn = 0;
            for (size_t c = 0; c < np; c++) dxn += dx[c] * dx[c];
            double *row = trace ? trace + 8 * (size_t)ntr : NULL;
            if (e_test < energy) {
                S rho_scale = 0;
                for (size_t c = 0; c < np; c++) rho_scale += dx[c] * (lambda * dx[c] + g[c]);
                const S rho = (energy - e_test) / rho_scale;
                const S lam_used = lambda;
                const S tm = (S)2.0 * rho - (S)1.0;
                S mul = (S)1.0 - tm * tm * tm;
                if (mul < (S)1.0 / (S)3.0) mul = (S)1.0 / (S)3.0;
                lambda *= mul;
                if (lambda < lam_min) lambda = lam_min;
                if (row) {
                    row[0] = iter; row[1] = 1; row[2] = (double)energy; row[3] = (double)rho; row[4] = (double)lambda;
                    row[5] = (double)lam_used; row[6] = (double)e_test; row[7] = sqrt((double)dxn);
                }
                ntr++;
                lambda_inc = inc_base;
                energy = e_test;
                hist[iter % 2] = energy;
                break;
            } else {
                if (row) {
                    row[0] = iter; row[1] = 0; row[2] = (double)energy; row[3] = 0; row[4] = (double)lambda;
                    row[5] = (double)lambda; row[6] = (double)e_test; row[7] = sqrt((double)dxn);
                }
                ntr++;
                if (lambda > lam_max) { status = 1; stop = 1; break; }
                lambda *= lambda_inc;
                lambda_inc = POW(lambda_inc, (S)1.5);
            }
        }
        if (stop) break;
        if (iter > 2) {
            const S maxf = hist[0] > hist[1] ? hist[0] : hist[1];
            if (FABS(energy - maxf) < tol_fun * energy) { status = 0; break; }
        }
        memcpy(cam15, camT, sizeof(S) * 15 * (size_t)N);
        memcpy(pts, ptsT, sizeof(S) * 3 * (size_t)M);
    }
    if (ntrials_out) *ntrials_out = ntr;
    free(fvec); free(Jc); free(Jp); free(dx); free(g); free(camT); free(ptsT);
    return status;
}

#undef EPS_PSI
