// ba_kernels.hip.h -- gfx950 kernels of the bundle-adjustment LM trial, templated on Scalar (double / float,
// src/BATypeUtils.h:6-7).  64-wide wavefronts; all cross-lane sums use a fixed order, so results are run-to-run
// reproducible.  HBM layout:
//   cam   T[15][N]   SoA  R row-major (9), T (3), f = K(0,0), k1, k2       (state, x and xTest copies)
//   pts   T[3][Ml]   SoA
//   meas  T[2][Kl], r T[2][Kl], Jc T[18][Kl] (2x9 row-major per obs), Jp T[6][Kl]   SoA over observations
//   rec   T[Kl][32]  AoS  per observation: Z (9x3 row-major), dinv of its point -- two 128-byte lines (fp64), gathered by the pair kernel
//   U0 T[6][Ml], gp T[3][Ml], dinv T[3][Ml], tvec T[3][Ml], tri T[6][Ml]            per point
//   S     T[Dp][Dp]  column-major, lower triangle + augmented rows D (rhs), D+1 (g_c), D+2 (scalars)
#ifndef BA_KERNELS_HIP_H
#define BA_KERNELS_HIP_H

#include <hip/hip_runtime.h>

#include "ba_mfma.hip.h"

#define BA_REC 32 /* Z (27), dinv of the point (3), pad (2): 256 B = two cache lines in fp64, one in fp32 */
#define BA_REC_DINV 27
#define BA_SLAB 96
#define BA_CHUNK 64 /* entries of one camera pair per chunk = per wavefront of the pair kernel */
#define BA_EPS_PSI 1e-15 /* src/Optimization/BAFunctor.h:159 */

__device__ __forceinline__ double tsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float tsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double tsin(double x) { return sin(x); }
__device__ __forceinline__ float tsin(float x) { return sinf(x); }
__device__ __forceinline__ double tcos(double x) { return cos(x); }
__device__ __forceinline__ float tcos(float x) { return cosf(x); }
#ifndef BA_SENTINEL_DEFINED
#define BA_SENTINEL_DEFINED
// A bit pattern no arithmetic produces (hardware NaNs are canonical): marks "not written yet" in a hand-off buffer.
template <typename T> __device__ __forceinline__ T ba_sentinel();
template <> __device__ __forceinline__ double ba_sentinel<double>() { return __hiloint2double(-1, -1); }
template <> __device__ __forceinline__ float ba_sentinel<float>() { return __int_as_float(-1); }
__device__ __forceinline__ bool ba_is_sentinel(double v) { return __double2hiint(v) == -1 && __double2loint(v) == -1; }
__device__ __forceinline__ bool ba_is_sentinel(float v) { return __float_as_int(v) == -1; }
#endif

template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }

// ---- block reductions (fixed order: shuffle tree inside a wave, then waves in index order) -----------------
template <typename T, bool MAX> __device__ __forceinline__ T wave_reduce(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_down(v, off, 64);
        v = MAX ? tmax(v, o) : v + o;
    }
    return v;
}

template <typename T, bool MAX> __device__ __forceinline__ T block_reduce(T v, T *lds /* >= blockDim/64 */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_reduce<T, MAX>(v);
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    T a = lds[0];
    for (int k = 1; k < nw; k++) a = MAX ? tmax(a, lds[k]) : a + lds[k];
    return a;
}

// ---- shared arithmetic of the point elimination (one source for the stand-alone kernels and for the fused linearisation: the
// same expressions, hence the same bits) ---------------------------------------------------------------------------------------
// The nine per-observation terms of a point's block: u[0..5] of B^T B (00 01 02 11 12 22) and u[6..8] of B^T r.
// (Every fused multiply-add of the three helpers is written out and contraction is off inside them: which product of a*b + c*d the
// compiler fuses is ITS choice per call site -- measured: the fused and the stand-alone kernels made different ones -- and leaving
// the fusing out altogether cost accuracy that the referee tests see (d1 = (u3 + lambda) - l10^2 d0 cancels).  So: one rounding per
// fma() below, the same bits whoever calls.)
__device__ __forceinline__ double ba_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float ba_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <typename T> __device__ __forceinline__ void ba_pt_terms(const T (&B)[6], T r0, T r1, T (&u)[9])
{
#pragma clang fp contract(off)
    u[0] = ba_fma(B[0], B[0], B[3] * B[3]);
    u[1] = ba_fma(B[0], B[1], B[3] * B[4]);
    u[2] = ba_fma(B[0], B[2], B[3] * B[5]);
    u[3] = ba_fma(B[1], B[1], B[4] * B[4]);
    u[4] = ba_fma(B[1], B[2], B[4] * B[5]);
    u[5] = ba_fma(B[2], B[2], B[5] * B[5]);
#pragma unroll
    for (int q = 0; q < 3; q++) u[6 + q] = ba_fma(B[q], r0, B[3 + q] * r1);
}
// U + lambda I = L D L^T (3 x 3, no pivoting), t = L^-1 g  (BacktrackLevMarqCholesky.h:274-282, point columns first)
template <typename T> struct ba_chol3_t { T l10, l20, l21, i0, i1, i2, t0, t1, t2; };
template <typename T> __device__ __forceinline__ ba_chol3_t<T> ba_chol3(T u0, T u1, T u2, T u3, T u4, T u5, T g0, T g1, T g2, T lambda)
{
#pragma clang fp contract(off)
    ba_chol3_t<T> c;
    const T d0 = u0 + lambda;
    c.l10 = u1 / d0; c.l20 = u2 / d0;
    const T d1 = ba_fma(-(c.l10 * c.l10), d0, u3 + lambda);
    c.l21 = ba_fma(-(c.l20 * c.l10), d0, u4) / d1;
    const T d2 = ba_fma(-(c.l21 * c.l21), d1, ba_fma(-(c.l20 * c.l20), d0, u5 + lambda));
    c.i0 = (T)1.0 / d0; c.i1 = (T)1.0 / d1; c.i2 = (T)1.0 / d2;
    c.t0 = g0; c.t1 = ba_fma(-c.l10, c.t0, g1); c.t2 = ba_fma(-c.l21, c.t1, ba_fma(-c.l20, c.t0, g2));
    return c;
}
// The observation's record: Z = A^T (B L^-T) (9 x 3 row-major), the point's 1 / D, pad -- one burst of 16-byte stores.
// The record (30 of its 32 scalars) leaves in one burst at the end: a record is two cache lines that only this thread writes, and
// stores trickling out between the Jacobian loads left them half-written in L2 for a microsecond -- evicted partial lines made 96 MB
// of HBM writes out of 58 MB at config 4 (rocprofv3 WRITE_SIZE).
template <typename T> __device__ __forceinline__ void ba_chol_record(const T (&A)[18], const T (&B)[6], T l10, T l20, T l21, T i0, T i1, T i2,
                                                                     T *__restrict__ rec_i)
{
#pragma clang fp contract(off)
    T Bt[6];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        Bt[3 * rr] = B[3 * rr];
        Bt[3 * rr + 1] = ba_fma(-l10, Bt[3 * rr], B[3 * rr + 1]);
        Bt[3 * rr + 2] = ba_fma(-l21, Bt[3 * rr + 1], ba_fma(-l20, Bt[3 * rr], B[3 * rr + 2]));
    }
    T z[BA_REC];
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const T a0 = A[c], a1 = A[9 + c];
        z[3 * c] = ba_fma(a0, Bt[0], a1 * Bt[3]); z[3 * c + 1] = ba_fma(a0, Bt[1], a1 * Bt[4]); z[3 * c + 2] = ba_fma(a0, Bt[2], a1 * Bt[5]);
    }
    z[BA_REC_DINV] = i0; z[BA_REC_DINV + 1] = i1; z[BA_REC_DINV + 2] = i2;
    z[BA_REC_DINV + 3] = 0; z[BA_REC_DINV + 4] = 0;
    typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
    vec_t *o = (vec_t *)rec_i;
    constexpr int VW = 16 / sizeof(T);
#pragma unroll
    for (int q = 0; q < BA_REC / VW; q++) {
        vec_t v;
#pragma unroll
        for (int u = 0; u < VW; u++) v[u] = z[VW * q + u];
        o[q] = v;
    }
}

// ---- K1/K2: residual (+ Jacobian) per observation ------------------------------------------------------------
// BAFunctor::E_pos (src/Optimization/BAFunctor.h:160-178) and dE_pos (:181-297) with poseDerivatives (:126-142),
// DistortionFunction (src/DistortionFunction.cpp:14-51), transformPointIntoCameraSpace (src/CameraMatrix.cpp:259-261).
// One thread per observation; camera / point parameters are gathered (observations are point-sorted, so a wave's
// point reads are near-contiguous and the 15N camera scalars stay in L2), everything else is a coalesced SoA stream.
//
// eb != nullptr: the workgroups' observation ranges are POINT-ALIGNED (eb[b] .. eb[b + 1], at most 256, whole points: built once on
// the host) instead of plain runs of 256 -- the same mapping for the residual-only evaluation, so the energy of a point x is the
// same bits whichever instantiation sums it.  That makes a point's observations neighbours inside ONE workgroup, and the
// linearisation can finish the point's part of the trial that follows it in the same pass (round 4, VERDICT r3 item 5: "never store
// J twice"):
//   FUSE >= 1  U0_j = sum B^T B and g_p = -sum B^T r per point (k_point_prep's work: Jp and r are not read back), the per-observation
//              terms meeting in LDS and summed by the point's first lane in observation order -- k_point_prep's order, its bits;
//   FUSE == 2  (CHOLESKY) the point's 3 x 3 LDL^T at the lambda the step control has just left in scal[], t, and every observation's
//              record Z_i -- k_elim_chol's work for the FIRST trial of the new outer iteration (Jc, Jp are not read back for it);
//              *fresh = 1 tells that trial's k_elim_chol to leave at once.  Rejected trials re-run k_elim_chol from the stored J.
template <typename T> struct ba_fuse_args {
    const int *eb;     // nullptr: plain runs of 256 observations
    const int *pt_ptr; // first observation of every point (+ end)
    const T *lam;
    T *U0, *gp, *rec, *dinv, *tvec, *tri;
    int *fresh;
};
// SOA = false (CHOLESKY): the camera blocks are kept in the AoS records JcA alone -- k_cam_gram gathers them by camera, k_elim_chol reads
// its own record -- and the 18 SoA streams Jc (144 of the 624 bytes this kernel writes per observation when fused) are not written.
template <typename T, bool JAC, int FUSE = 0, bool SOA = true>
__global__ __launch_bounds__(256) void k_eval(int K, int N, int Ml, const T *__restrict__ cam, const T *__restrict__ pts,
                                              const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                                              const T *__restrict__ meas, T tau2, T *__restrict__ r, T *__restrict__ Jc,
                                              T *__restrict__ Jp, T *__restrict__ JcA /* [K][20] AoS copy: A (18), r (2) */,
                                              T *__restrict__ partial, const int *__restrict__ go = nullptr,
                                              T *__restrict__ commit_cam = nullptr, T *__restrict__ commit_pts = nullptr,
                                              ba_fuse_args<T> fa = ba_fuse_args<T>{})
{
    static_assert(FUSE == 0 || JAC, "the fused point part belongs to the linearisation");
    __shared__ T red[4];
    if (go && *go == 0) return; // device-side LM control: the trial in front of this linearisation was rejected (uniform)
    const int b0 = fa.eb ? fa.eb[blockIdx.x] : (int)blockIdx.x * 256;
    const int b1 = fa.eb ? fa.eb[blockIdx.x + 1] : (b0 + 256 < K ? b0 + 256 : K);
    const int i = b0 + threadIdx.x;
    const bool valid = i < b1;
    if (commit_cam) { // x = xTest (BacktrackLevMarqQRChol.h:428) rides on the linearisation at xTest: cam / pts ARE xTest here
        const int ncam = 15 * N, ntot = ncam + 3 * Ml;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < ntot; q += gridDim.x * 256) {
            if (q < ncam) commit_cam[q] = cam[q];
            else commit_pts[q - ncam] = pts[q - ncam];
        }
    }
    T e2 = 0;
    T Aj[FUSE == 2 ? 18 : 1], Bj[FUSE > 0 ? 6 : 1], ej[2] = {0, 0};
    int pj = 0;
    if (valid) {
        const int ci = obs_cam[i];
        pj = obs_pt[i];
        T c[15];
#pragma unroll
        for (int k = 0; k < 15; k++) c[k] = cam[(size_t)k * N + ci];
        const T X0 = pts[pj], X1 = pts[(size_t)Ml + pj], X2 = pts[2 * (size_t)Ml + pj];
        const T RX0 = c[0] * X0 + c[1] * X1 + c[2] * X2;
        const T RX1 = c[3] * X0 + c[4] * X1 + c[5] * X2;
        const T RX2 = c[6] * X0 + c[7] * X1 + c[8] * X2;
        const T XX0 = RX0 + c[9], XX1 = RX1 + c[10], XX2 = RX2 + c[11];
        const T xu0 = XX0 / XX2, xu1 = XX1 / XX2;
        const T r2u = xu0 * xu0 + xu1 * xu1, r4u = r2u * r2u;
        const T f = c[12], k1 = c[13], k2 = c[14];
        const T kr = 1 + k1 * r2u + k2 * r4u;
        const T xd0 = kr * xu0, xd1 = kr * xu1;
        const T r0 = f * xd0 - meas[i], r1 = f * xd1 - meas[(size_t)K + i];
        const T r2 = r0 * r0 + r1 * r1;
        // psi (BAFunctor.h:147)
        const T psi = (r2 < tau2) ? r2 * ((T)2.0 - r2 / tau2) / (T)4.0 : tau2 / (T)4.0;
        const T sqrt_psi = tsqrt(psi);
        const T nr = tsqrt(r2);
        const T rnorm_r = (T)1.0 / tmax((T)BA_EPS_PSI, nr);
        const T e0 = r0 * sqrt_psi * rnorm_r, e1 = r1 * sqrt_psi * rnorm_r;
        e2 = e0 * e0 + e1 * e1;
        if (JAC) {
            ej[0] = e0; ej[1] = e1;
            r[i] = e0;
            r[(size_t)K + i] = e1;
            JcA[(size_t)i * 20 + 18] = e0;
            JcA[(size_t)i * 20 + 19] = e1;
            // -[XX - T]x  (poseDerivatives, BAFunctor.h:131-133; XX - T is formed as the reference forms it)
            const T v0 = XX0 - c[9], v1 = XX1 - c[10], v2 = XX2 - c[11];
            const T mJ[9] = {0, v2, -v1, -v2, 0, v0, v1, -v0, 0};
            const T a00 = (T)1.0 / XX2, a02 = -XX0 / (XX2 * XX2), a12 = -XX1 / (XX2 * XX2);
            const T dkr = 2 * k1 + 4 * k2 * r2u;
            const T d00 = kr + xu0 * xu0 * dkr, d01 = xu0 * xu1 * dkr, d11 = kr + xu1 * xu1 * dkr;
            const T p00 = f * d00, p01 = f * d01, p11 = f * d11;
            T dpX[6];
            dpX[0] = p00 * a00; dpX[1] = p01 * a00; dpX[2] = p00 * a02 + p01 * a12;
            dpX[3] = p01 * a00; dpX[4] = p11 * a00; dpX[5] = p01 * a02 + p11 * a12;
            // outer derivative of the robustified residual (BAFunctor.h:227-242)
            const T tw = (T)1.0 - r2 / tau2;
            const T W = tw > (T)0 ? tw : (T)0;
            const T rsqrt_psi = (T)1.0 / tmax((T)BA_EPS_PSI, sqrt_psi);
            const T rcp_r2 = (T)1.0 / tmax((T)BA_EPS_PSI, r2);
            const T rr00 = r0 * r0 * rnorm_r, rr01 = r0 * r1 * rnorm_r, rr11 = r1 * r1 * rnorm_r;
            const T c1 = W / (T)2.0 * rsqrt_psi, c2 = sqrt_psi * rcp_r2;
            const T o00 = c1 * rr00 + c2 * (nr - rr00);
            const T o01 = c1 * rr01 + c2 * ((T)0 - rr01);
            const T o11 = c1 * rr11 + c2 * (nr - rr11);
            T Jb[24];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const T *d = dpX + 3 * rr;
                T *o = Jb + 12 * rr;
                o[0] = d[0]; o[1] = d[1]; o[2] = d[2];
#pragma unroll
                for (int q = 0; q < 3; q++) o[3 + q] = d[0] * mJ[q] + d[1] * mJ[3 + q] + d[2] * mJ[6 + q];
#pragma unroll
                for (int q = 0; q < 3; q++) o[9 + q] = d[0] * c[q] + d[1] * c[3 + q] + d[2] * c[6 + q];
            }
            Jb[6] = xd0; Jb[18] = xd1;
            Jb[7] = f * (xu0 * r2u); Jb[8] = f * (xu0 * r4u);
            Jb[19] = f * (xu1 * r2u); Jb[20] = f * (xu1 * r4u);
#pragma unroll
            for (int q = 0; q < 12; q++) {
                const T t0 = o00 * Jb[q] + o01 * Jb[12 + q];
                const T t1 = o01 * Jb[q] + o11 * Jb[12 + q];
                if (q < 9) {
                    if (SOA) {
                        Jc[(size_t)q * K + i] = t0;
                        Jc[(size_t)(9 + q) * K + i] = t1;
                    }
                    JcA[(size_t)i * 20 + q] = t0; // gathered by camera in k_cam_gram: one 160-byte record per observation
                    JcA[(size_t)i * 20 + 9 + q] = t1;
                    if (FUSE == 2) { Aj[q] = t0; Aj[9 + q] = t1; }
                } else {
                    Jp[(size_t)(q - 9) * K + i] = t0;
                    Jp[(size_t)(q - 6) * K + i] = t1;
                    if (FUSE > 0) { Bj[q - 9] = t0; Bj[q - 6] = t1; }
                }
            }
        }
    }
    e2 = block_reduce<T, false>(e2, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = e2;
    if constexpr (FUSE > 0) {
        // the point part: a point's observations are threads tl .. tl + k - 1 of this workgroup (point-aligned ranges)
        __shared__ T cu[9][256];
        const int tl = threadIdx.x;
        {
            T u[9];
#pragma unroll
            for (int q = 0; q < 9; q++) u[q] = 0;
            if (valid) ba_pt_terms<T>(Bj, ej[0], ej[1], u);
#pragma unroll
            for (int q = 0; q < 9; q++) cu[q][tl] = u[q];
        }
        __syncthreads();
        const int pfirst = valid ? fa.pt_ptr[pj] : 0;
        __shared__ T pf[6][256];
        if (valid && i == pfirst) {
            const int kk = fa.pt_ptr[pj + 1] - pfirst;
            T U[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            for (int s = 0; s < kk; s++) { // observation order: k_point_prep's order of additions
#pragma unroll
                for (int q = 0; q < 6; q++) U[q] += cu[q][tl + s];
#pragma unroll
                for (int q = 0; q < 3; q++) g[q] -= cu[6 + q][tl + s];
            }
#pragma unroll
            for (int q = 0; q < 6; q++) fa.U0[(size_t)q * Ml + pj] = U[q];
#pragma unroll
            for (int q = 0; q < 3; q++) fa.gp[(size_t)q * Ml + pj] = g[q];
            if (FUSE == 2) {
                const ba_chol3_t<T> c3 = ba_chol3<T>(U[0], U[1], U[2], U[3], U[4], U[5], g[0], g[1], g[2], *fa.lam);
                const size_t M = (size_t)Ml;
                fa.dinv[pj] = c3.i0; fa.dinv[M + pj] = c3.i1; fa.dinv[2 * M + pj] = c3.i2;
                fa.tvec[pj] = c3.t0; fa.tvec[M + pj] = c3.t1; fa.tvec[2 * M + pj] = c3.t2;
                fa.tri[pj] = 1; fa.tri[M + pj] = c3.l10; fa.tri[2 * M + pj] = c3.l20;
                fa.tri[3 * M + pj] = 1; fa.tri[4 * M + pj] = c3.l21; fa.tri[5 * M + pj] = 1;
                pf[0][tl] = c3.l10; pf[1][tl] = c3.l20; pf[2][tl] = c3.l21; pf[3][tl] = c3.i0; pf[4][tl] = c3.i1; pf[5][tl] = c3.i2;
            }
        }
        if constexpr (FUSE == 2) {
            __syncthreads();
            if (valid) {
                const int fl = pfirst - b0; // the point's first lane
                ba_chol_record<T>(Aj, Bj, pf[0][fl], pf[1][fl], pf[2][fl], pf[3][fl], pf[4][fl], pf[5][fl], fa.rec + (size_t)i * BA_REC);
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) *fa.fresh = 1;
        }
    }
}

// Utils::showErrorStatistics / showObjective (src/Utils.h:10-68): 4 partial sums per block.
template <typename T>
__global__ __launch_bounds__(256) void k_stats(int K, int N, int Ml, const T *__restrict__ cam, const T *__restrict__ pts,
                                               const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                                               const T *__restrict__ meas, T tau, T *__restrict__ partial /* [4][grid] */)
{
    __shared__ T red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    T v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (i < K) {
        const int ci = obs_cam[i], pj = obs_pt[i];
        T c[15];
#pragma unroll
        for (int k = 0; k < 15; k++) c[k] = cam[(size_t)k * N + ci];
        const T X0 = pts[pj], X1 = pts[(size_t)Ml + pj], X2 = pts[2 * (size_t)Ml + pj];
        const T XX0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
        const T XX1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
        const T XX2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
        const T xu0 = XX0 / XX2, xu1 = XX1 / XX2;
        const T r2u = xu0 * xu0 + xu1 * xu1;
        const T kr = 1 + c[13] * r2u + c[14] * r2u * r2u;
        const T d0 = c[12] * (kr * xu0) - meas[i], d1 = c[12] * (kr * xu1) - meas[(size_t)K + i];
        const T err = tsqrt(d0 * d0 + d1 * d1);
        const T tau2 = tau * tau, tau4 = tau2 * tau2;
        v0 = err;
        if (err <= tau) { v1 = err; v2 = 1; }
        const T q2 = err, q4 = q2 * q2; // sic: the norm, not its square, goes into Utils::psi (Utils.h:61-62)
        v3 = (q2 < tau2) ? q2 * ((T)3.0 - (T)3.0 * q2 / tau2 + q4 / tau4) / (T)6.0 : tau2 / (T)6.0;
    }
    v0 = block_reduce<T, false>(v0, red);
    v1 = block_reduce<T, false>(v1, red);
    v2 = block_reduce<T, false>(v2, red);
    v3 = block_reduce<T, false>(v3, red);
    if (threadIdx.x == 0) {
        const size_t g = gridDim.x;
        partial[blockIdx.x] = v0; partial[g + blockIdx.x] = v1; partial[2 * g + blockIdx.x] = v2; partial[3 * g + blockIdx.x] = v3;
    }
}

// ---- second stage of every scalar reduction: up to 8 (array, count, op) jobs, one block each ----------------
struct ba_red_job { const void *src; int n; int op; int dst; }; // op 0 sum, 1 max
struct ba_red_jobs { ba_red_job j[8]; };

// stamp != nullptr: block 0 leaves the device's wall clock there when it is done (also when `go` turns the launch into a no-op):
// the last kernel of an LM iteration's control segment marks its end for k_lm_control's per-trial timing.
// one job by one 256-thread block: fixed order (thread k, k + 256, ...; tree inside a wave; waves in index order)
template <typename T> __device__ __forceinline__ void ba_reduce_job(const ba_red_job &jb, T *__restrict__ scal, T *red)
{
    const T *src = (const T *)jb.src;
    T a = 0;
    if (jb.op == 0) {
        for (int k = threadIdx.x; k < jb.n; k += 256) a += src[k];
        a = block_reduce<T, false>(a, red);
    } else {
        for (int k = threadIdx.x; k < jb.n; k += 256) a = tmax(a, src[k]);
        a = block_reduce<T, true>(a, red);
    }
    if (threadIdx.x == 0) scal[jb.dst] = a;
}

template <typename T> __global__ __launch_bounds__(256) void k_reduce_scalars(ba_red_jobs jobs, T *__restrict__ scal, const int *__restrict__ go = nullptr,
                                                                              long long *__restrict__ stamp = nullptr)
{
    __shared__ T red[4];
    if (!(go && *go == 0)) ba_reduce_job<T>(jobs.j[blockIdx.x], scal, red);
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp = (long long)wall_clock64();
}

// ---- K3 (point part): U0_j = sum B^T B, g_p = -sum B^T r per point, once per outer iteration ------------------
// JtRes and the squared column norms of the point columns (src/Eigen_ext/BacktrackLevMarqQRChol.h:267-274).
template <typename T>
__device__ __forceinline__ void ba_point_prep_block(int bid, int Ml, int K, const int *__restrict__ pt_ptr, const T *__restrict__ Jp,
                                                    const T *__restrict__ r, T *__restrict__ U0, T *__restrict__ gp, T *__restrict__ partial_dmax)
{
    __shared__ T red[4];
    const int j = bid * 256 + threadIdx.x;
    T dm = 0;
    if (j < Ml) {
        T U[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        const int e = pt_ptr[j + 1];
        for (int i = pt_ptr[j]; i < e; i++) {
            T B[6];
#pragma unroll
            for (int q = 0; q < 6; q++) B[q] = Jp[(size_t)q * K + i];
            const T r0 = r[i], r1 = r[(size_t)K + i];
            T u[9];
            ba_pt_terms<T>(B, r0, r1, u); // (shared with the fused linearisation, k_eval<T, true, FUSE>: same terms, same order)
#pragma unroll
            for (int q = 0; q < 6; q++) U[q] += u[q];
#pragma unroll
            for (int q = 0; q < 3; q++) g[q] -= u[6 + q];
        }
#pragma unroll
        for (int q = 0; q < 6; q++) U0[(size_t)q * Ml + j] = U[q];
#pragma unroll
        for (int q = 0; q < 3; q++) gp[(size_t)q * Ml + j] = g[q];
        dm = tmax(U[0], tmax(U[3], U[5]));
    }
    dm = block_reduce<T, true>(dm, red);
    if (threadIdx.x == 0) partial_dmax[bid] = dm;
}
template <typename T>
__global__ __launch_bounds__(256) void k_point_prep(int Ml, int K, const int *__restrict__ pt_ptr, const T *__restrict__ Jp,
                                                    const T *__restrict__ r, T *__restrict__ U0, T *__restrict__ gp,
                                                    T *__restrict__ partial_dmax, const int *__restrict__ go = nullptr)
{
    if (go && *go == 0) return;
    ba_point_prep_block<T>(blockIdx.x, Ml, K, pt_ptr, Jp, r, U0, gp, partial_dmax);
}

// ---- K3 (camera part): V_aa = sum A^T A (9x9) and g_c = -sum A^T r per camera, once per outer iteration -------
// One 32-lane group per chunk of <= 32 observations of one camera, lane = observation: every lane gathers its own 2x9
// block (20 independent loads in flight), forms the 45 unique products of A^T A and the 9 of A^T r in registers, and the
// 54 values are summed over the 32 lanes through LDS (two passes of 27 rows x 32 lanes, each row summed by one lane in
// lane order -> deterministic).  A lane-per-output mapping walked the 32 observations one after the other and was
// latency-bound (85 us for 226 k observations).
template <typename T>
__device__ __forceinline__ void ba_cam_gram_block(int bid, int ndchunks, int K, const int *__restrict__ dchunk_ptr, const int *__restrict__ cam_obs,
                                                  const T *__restrict__ JcA, T *__restrict__ dslab)
{
    __shared__ T xch[8][27][33];
    const int gl = threadIdx.x >> 5, g = bid * 8 + gl, sub = threadIdx.x & 31;
    const bool gok = g < ndchunks;
    const int e0 = gok ? dchunk_ptr[g] : 0, len = gok ? dchunk_ptr[g + 1] - e0 : 0;
    T v[54];
#pragma unroll
    for (int q = 0; q < 54; q++) v[q] = 0;
    if (sub < len) {
        const T *rec = JcA + (size_t)cam_obs[e0 + sub] * 20;
        T a0[9], a1[9];
#pragma unroll
        for (int c = 0; c < 9; c++) { a0[c] = rec[c]; a1[c] = rec[9 + c]; }
        const T r0 = rec[18], r1 = rec[19];
        int q = 0;
#pragma unroll
        for (int c = 0; c < 9; c++)
#pragma unroll
            for (int c2 = 0; c2 <= c; c2++) v[q++] = a0[c] * a0[c2] + a1[c] * a1[c2]; // lower triangle, row-major
#pragma unroll
        for (int c = 0; c < 9; c++) v[45 + c] = -(a0[c] * r0 + a1[c] * r1);
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 27; q++) xch[gl][q][sub] = v[27 * pass + q];
        __syncthreads();
        if (gok && sub < 27) {
            T a = 0;
#pragma unroll
            for (int l = 0; l < 32; l++) a += xch[gl][sub][l];
            dslab[(size_t)g * BA_SLAB + 27 * pass + sub] = a;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_cam_gram(int ndchunks, int K, const int *__restrict__ dchunk_ptr,
                                                  const int *__restrict__ cam_obs, const T *__restrict__ JcA,
                                                  T *__restrict__ dslab, const int *__restrict__ go = nullptr)
{
    if (go && *go == 0) return;
    ba_cam_gram_block<T>(blockIdx.x, ndchunks, K, dchunk_ptr, cam_obs, JcA, dslab);
}
// Both in ONE launch (the first gM workgroups take the points, the rest the cameras' chunks): the two are independent, and for a
// problem whose point part is a single round of workgroups (config 4: 255) neither fills the chip by itself -- 10.6 + 13.4 us as two
// launches.  (Every workgroup reserves the camera part's 57 KB of LDS, so a larger point part would lose its occupancy: the host
// keeps the two launches there.)
template <typename T>
__global__ __launch_bounds__(256) void k_grad_prep(int gM, int Ml, int K, const int *__restrict__ pt_ptr, const T *__restrict__ Jp, const T *__restrict__ r,
                                                   T *__restrict__ U0, T *__restrict__ gp, T *__restrict__ partial_dmax, int ndchunks,
                                                   const int *__restrict__ dchunk_ptr, const int *__restrict__ cam_obs, const T *__restrict__ JcA,
                                                   T *__restrict__ dslab, const int *__restrict__ go = nullptr)
{
    if (go && *go == 0) return;
    if ((int)blockIdx.x < gM) ba_point_prep_block<T>(blockIdx.x, Ml, K, pt_ptr, Jp, r, U0, gp, partial_dmax); // (uniform per workgroup)
    else ba_cam_gram_block<T>(blockIdx.x - gM, ndchunks, K, dchunk_ptr, cam_obs, JcA, dslab);
}

// tail.src != nullptr: one more block at the end of the grid sums the energy partials of k_eval (the reduction that closes a
// linearisation; a launch of its own otherwise) and leaves the wall-clock stamp that ends the control segment (k_lm_control).
template <typename T>
__global__ __launch_bounds__(256) void k_cam_gram_reduce(int N, const int *__restrict__ cam_dchunk_ptr,
                                                         const T *__restrict__ dslab, T *__restrict__ V /* [N][81] */,
                                                         T *__restrict__ gc /* [9N] */, const int *__restrict__ go = nullptr,
                                                         ba_red_job tail = ba_red_job{nullptr, 0, 0, 0}, T *__restrict__ scal = nullptr,
                                                         long long *__restrict__ stamp = nullptr)
{
    __shared__ T red[4];
    if (tail.src && blockIdx.x == gridDim.x - 1) {
        if (!(go && *go == 0)) ba_reduce_job<T>(tail, scal, red);
        if (stamp && threadIdx.x == 0) *stamp = (long long)wall_clock64();
        return;
    }
    if (go && *go == 0) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int a = idx / BA_SLAB, e = idx - a * BA_SLAB;
    if (a >= N || e >= 54) return;
    // four interleaved partial sums in a fixed order; sixteen loads are issued before the first of their additions (the additions
    // and their order are those of a four-at-a-time loop -- the same bits -- but a camera of problem-21 has 54 chunks, and four loads
    // per L2 round trip made this launch a 14-trip latency chain: 9.1 us)
    T s4[4] = {0, 0, 0, 0};
    const int c0 = cam_dchunk_ptr[a], c1 = cam_dchunk_ptr[a + 1];
    int c = c0;
    for (; c + 15 < c1; c += 16) {
        T x[16];
#pragma unroll
        for (int u = 0; u < 16; u++) x[u] = dslab[(size_t)(c + u) * BA_SLAB + e];
#pragma unroll
        for (int u = 0; u < 16; u++) s4[u & 3] += x[u];
    }
    for (; c + 3 < c1; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) s4[u] += dslab[(size_t)(c + u) * BA_SLAB + e];
    }
    for (; c < c1; c++) s4[0] += dslab[(size_t)c * BA_SLAB + e];
    const T s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if (e < 45) {
        int rr = 0;
        while ((rr + 1) * (rr + 2) / 2 <= e) rr++;
        const int cc = e - rr * (rr + 1) / 2;
        V[(size_t)a * 81 + 9 * rr + cc] = s;
        V[(size_t)a * 81 + 9 * cc + rr] = s;
    } else
        gc[9 * a + e - 45] = s;
}

// diagonal of J_c^T J_c (squared column norms of the camera columns, BacktrackLevMarqQRChol.h:270-274)
template <typename T> __global__ __launch_bounds__(256) void k_vdiag(int N, const T *__restrict__ V, T *__restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < 9 * N) out[c] = V[(size_t)(c / 9) * 81 + 10 * (c % 9)];
}

// ---- K4 (CHOLESKY): eliminate the 3x3 point block of J^T J + lambda I ----------------------------------------
// LDL^T of the whole matrix with the point columns first (src/Eigen_ext/BacktrackLevMarqCholesky.h:274-282):
// U_j + lambda I = L D L^T;  Z_i = A_i^T (B_i L^-T) = W_i L^-T;  t = L^-1 g_p.  One thread per observation (each
// recomputes its point's 3x3 LDL^T: 20 flops, cheaper than a second launch); the first observation of a point also
// writes the per-point factors.
template <typename T>
__global__ __launch_bounds__(256) void k_elim_chol(int K, int Ml, const int *__restrict__ obs_pt, const int *__restrict__ pt_ptr,
                                                   const T *__restrict__ JcA /* [K][20] */, const T *__restrict__ Jp, const T *__restrict__ U0,
                                                   const T *__restrict__ gp, const T *__restrict__ lam, T *__restrict__ rec,
                                                   T *__restrict__ dinv, T *__restrict__ tvec, T *__restrict__ tri,
                                                   const int *__restrict__ fresh = nullptr /* != 0: the fused linearisation has left this trial's records */)
{
    if (fresh && *fresh != 0) return; // (uniform)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const T lambda = *lam;
    const int j = obs_pt[i];
    const size_t M = (size_t)Ml;
    const ba_chol3_t<T> c3 = ba_chol3<T>(U0[j], U0[M + j], U0[2 * M + j], U0[3 * M + j], U0[4 * M + j], U0[5 * M + j],
                                         gp[j], gp[M + j], gp[2 * M + j], lambda);
    if (i == pt_ptr[j]) {
        dinv[j] = c3.i0; dinv[M + j] = c3.i1; dinv[2 * M + j] = c3.i2;
        tvec[j] = c3.t0; tvec[M + j] = c3.t1; tvec[2 * M + j] = c3.t2;
        tri[j] = 1; tri[M + j] = c3.l10; tri[2 * M + j] = c3.l20;
        tri[3 * M + j] = 1; tri[4 * M + j] = c3.l21; tri[5 * M + j] = 1;
    }
    T A[18], B[6];
#pragma unroll
    for (int q = 0; q < 6; q++) B[q] = Jp[(size_t)q * K + i];
    { // the observation's own AoS record (A: 18 scalars, then its residual): 16-byte loads of the 160-byte (fp32: 80) record
        typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int VW = 16 / sizeof(T);
        const vec_t *src = (const vec_t *)(JcA + (size_t)i * 20);
#pragma unroll
        for (int q = 0; q < 20 / VW; q++) {
            const vec_t v = src[q];
#pragma unroll
            for (int u = 0; u < VW; u++)
                if (VW * q + u < 18) A[VW * q + u] = v[u];
        }
    }
    ba_chol_record<T>(A, B, c3.l10, c3.l20, c3.l21, c3.i0, c3.i1, c3.i2, rec + (size_t)i * BA_REC);
}

// ---- K4 (QRCHOL / QRKIT left block): Householder QR of [sqrt(lambda) I3 ; (Jp)_j] per point ------------------
// The block BlockDiagonalSparseQR factors (src/Optimization/BAFunctor.h:99-105, BAFunctor.cpp:64-68,
// src/Eigen_ext/BacktrackLevMarqQRChol.h:291-319), rows permuted so that the three sqrt(lambda) rows come first
// (R and Q1 are unique up to row / column signs, which cancel in S, the reduced rhs and dx), unpivoted.
//
// LPP lanes work on one point (8 points per wavefront at LPP = 8): lane g of the group holds observations
// g, g + LPP, ... of the point (up to 4 per lane, 2x3 Jacobian rows each) in registers, so the three Householder
// reflectors, the thin Q1 = H0 H1 H2 [I3;0] and q1 = Q1^T [0;r] need no memory round trips -- only nine butterfly sums
// over the LPP lanes.  A thread-per-point loop over observation rows in memory was latency-bound (150 us on 11 k points).
// The lanes then turn their observations' Q1 rows into Z_i = R12_i^T = A_i^T Q1_i and zt_i = Z_i t.
template <typename T, int LPP> __device__ __forceinline__ T group_sum(T v)
{
#pragma unroll
    for (int off = LPP / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LPP);
    return v;
}

template <typename T, int LPP, int SL> // SL observations per lane: points with up to SL * LPP observations
__global__ __launch_bounds__(256) void k_elim_qr(int npts, const int *__restrict__ pt_list, int Ml, int K, const int *__restrict__ pt_ptr,
                                                 const T *__restrict__ Jc, const T *__restrict__ Jp, const T *__restrict__ r,
                                                 const T *__restrict__ lam, T *__restrict__ rec, T *__restrict__ dinv,
                                                 T *__restrict__ tvec, T *__restrict__ tri, const int *__restrict__ go = nullptr,
                                                 T *__restrict__ q1obs = nullptr /* [K][6]: thin Q rows per observation (QRKIT) */,
                                                 T *__restrict__ q1lam = nullptr /* [Ml][9]: thin Q rows of the lambda rows */,
                                                 int *__restrict__ pperm = nullptr /* [Ml]: column permutation p0 | p1 << 2 | p2 << 4 */)
{
    if (go && *go == 0) return; // (MOREQR's outer factorisation is part of the conditional linearisation)
    // the points of one track-length bucket (pt_list; the host buckets them so that a short track does not occupy the lanes
    // of the longest one)
    const int gid = (blockIdx.x * 256 + threadIdx.x) / LPP, lg = threadIdx.x % LPP;
    const int j = pt_list[gid < npts ? gid : npts - 1]; // idle groups shadow the last point (no early exit: shuffles need every lane)
    const int b = pt_ptr[j], k = pt_ptr[j + 1] - b;
    const T sl = tsqrt(*lam);
    T V[SL][6], Q[SL][6];
    bool ok[SL];
#pragma unroll
    for (int s = 0; s < SL; s++) {
        ok[s] = s * LPP + lg < k;
        const int i = b + s * LPP + lg;
#pragma unroll
        for (int q = 0; q < 6; q++) { V[s][q] = ok[s] ? Jp[(size_t)q * K + i] : (T)0; Q[s][q] = 0; }
    }
    // Column pivoting like the reference's dense block solver (ColPivHouseholderQR, BAFunctor.h:99,104): every step takes the
    // remaining column of largest norm.  The remaining columns all carry the same sqrt(lambda) in their own (untouched) lambda
    // row, so the comparison is between the observation parts; the first maximum wins a tie.  A step's choice is the same in all
    // lanes of a point's group (group sums), so the swaps are selects on registers; perm[c] = original index of the column in
    // position c, undone in k_backsub (colsPermutation(), BacktrackLevMarqQRChol.h:360).
    T R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, tau[3];
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        T nrm[3] = {0, 0, 0};
#pragma unroll
        for (int c2 = c; c2 < 3; c2++) {
#pragma unroll
            for (int s = 0; s < SL; s++) nrm[c2] += V[s][c2] * V[s][c2] + V[s][3 + c2] * V[s][3 + c2];
            nrm[c2] = group_sum<T, LPP>(nrm[c2]);
        }
        int best = c;
#pragma unroll
        for (int c2 = c + 1; c2 < 3; c2++)
            if (nrm[c2] > nrm[best]) best = c2;
#pragma unroll
        for (int c2 = c + 1; c2 < 3; c2++) { // (best is lane-varying between the groups of a wave: swap by selects)
            const bool sw = best == c2;
#pragma unroll
            for (int s = 0; s < SL; s++) {
                const T a0 = V[s][c], a1 = V[s][3 + c];
                V[s][c] = sw ? V[s][c2] : a0; V[s][c2] = sw ? a0 : V[s][c2];
                V[s][3 + c] = sw ? V[s][3 + c2] : a1; V[s][3 + c2] = sw ? a1 : V[s][3 + c2];
            }
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
                if (rr < c) { const T t0 = R[rr][c]; R[rr][c] = sw ? R[rr][c2] : t0; R[rr][c2] = sw ? t0 : R[rr][c2]; }
            const int p0 = perm[c];
            perm[c] = sw ? perm[c2] : p0; perm[c2] = sw ? p0 : perm[c2];
            const T n0 = nrm[c];
            nrm[c] = sw ? nrm[c2] : n0; nrm[c2] = sw ? n0 : nrm[c2];
        }
        const T xn = nrm[c];
        const T alpha = sl;                        // the lambda row c is untouched by the earlier reflectors
        const T beta = -tsqrt(alpha * alpha + xn); // alpha >= 0
        // beta == 0: a zero column, possible only with lambda = 0 (MOREQR stage 1, point without observations):
        // identity reflector, zero diagonal entry of R
        const bool zc = beta == (T)0;
        tau[c] = zc ? (T)0 : (beta - alpha) / beta;
        const T sc = zc ? (T)0 : (T)1.0 / (alpha - beta);
#pragma unroll
        for (int s = 0; s < SL; s++) { V[s][c] *= sc; V[s][3 + c] *= sc; }
        R[c][c] = beta;
        T w[3] = {0, 0, 0};
#pragma unroll
        for (int c2 = c + 1; c2 < 3; c2++) {
#pragma unroll
            for (int s = 0; s < SL; s++) w[c2] += V[s][c] * V[s][c2] + V[s][3 + c] * V[s][3 + c2];
            w[c2] = group_sum<T, LPP>(w[c2]) * tau[c]; // the pivot-row entry of column c2 is still zero
            R[c][c2] = -w[c2];
#pragma unroll
            for (int s = 0; s < SL; s++) { V[s][c2] -= V[s][c] * w[c2]; V[s][3 + c2] -= V[s][3 + c] * w[c2]; }
        }
    }
    // thin Q1 = H0 H1 H2 [I3; 0]; Ql = its three lambda rows (identical in every lane)
    T Ql[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int h = 2; h >= 0; h--) {
        T w[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            T a = 0;
#pragma unroll
            for (int s = 0; s < SL; s++) a += V[s][h] * Q[s][c] + V[s][3 + h] * Q[s][3 + c];
            w[c] = (Ql[h][c] + group_sum<T, LPP>(a)) * tau[h];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Ql[h][c] -= w[c];
#pragma unroll
            for (int s = 0; s < SL; s++) { Q[s][c] -= V[s][h] * w[c]; Q[s][3 + c] -= V[s][3 + h] * w[c]; }
        }
    }
    // t = -Q1^T [0; r]
    T q1[3] = {0, 0, 0};
#pragma unroll
    for (int s = 0; s < SL; s++) {
        const int i = b + s * LPP + lg;
        const T r0 = ok[s] ? r[i] : (T)0, r1 = ok[s] ? r[(size_t)K + i] : (T)0;
#pragma unroll
        for (int c = 0; c < 3; c++) q1[c] += Q[s][c] * r0 + Q[s][3 + c] * r1;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) q1[c] = -group_sum<T, LPP>(q1[c]);
    if (gid < npts && lg == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { dinv[(size_t)c * Ml + j] = 1; tvec[(size_t)c * Ml + j] = q1[c]; }
        tri[j] = R[0][0]; tri[(size_t)Ml + j] = R[0][1]; tri[2 * (size_t)Ml + j] = R[0][2];
        tri[3 * (size_t)Ml + j] = R[1][1]; tri[4 * (size_t)Ml + j] = R[1][2]; tri[5 * (size_t)Ml + j] = R[2][2];
        if (pperm) pperm[j] = perm[0] | (perm[1] << 2) | (perm[2] << 4);
        if (q1lam) {
#pragma unroll
            for (int h = 0; h < 3; h++)
#pragma unroll
                for (int c = 0; c < 3; c++) q1lam[9 * (size_t)j + 3 * h + c] = Ql[h][c];
        }
    }
    if (q1obs && gid < npts) {
#pragma unroll
        for (int s = 0; s < SL; s++)
            if (ok[s]) {
#pragma unroll
                for (int q = 0; q < 6; q++) q1obs[6 * (size_t)(b + s * LPP + lg) + q] = Q[s][q];
            }
    }
    // Z_i = A_i^T Q1_i (9x3)
    if (gid < npts) {
#pragma unroll
        for (int s = 0; s < SL; s++) {
            if (!ok[s]) continue;
            const int i = b + s * LPP + lg;
            T *o = rec + (size_t)i * BA_REC;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const T a0 = Jc[(size_t)c * K + i], a1 = Jc[(size_t)(9 + c) * K + i];
                const T z0 = a0 * Q[s][0] + a1 * Q[s][3], z1 = a0 * Q[s][1] + a1 * Q[s][4], z2 = a0 * Q[s][2] + a1 * Q[s][5];
                o[3 * c] = z0; o[3 * c + 1] = z1; o[3 * c + 2] = z2;
            }
            o[BA_REC_DINV] = 1; o[BA_REC_DINV + 1] = 1; o[BA_REC_DINV + 2] = 1;
        }
    }
}

// MOREQR stage 2 (src/Eigen_ext/BacktrackLevMarqMore.h:297-345: the QR of [R ; sqrt(lambda) I] of every trial).  Stage 1
// (:288, m_solver.compute(J), once per outer iteration) is k_elim_qr with lambda = 0, which leaves R1_j, -q1_j and
// R12_i^T in (tri0, tvec0, rec0).  Per point the 6x3 block [sqrt(lambda) I3 ; R1_j] is factored again (three
// Householder reflectors, lambda rows first); with QR = the 3x3 block of its thin Q that multiplies the R1 rows:
// Z_i = R12_i^T QR, t = QR^T (-q1), tri = Rt1.  One thread per observation (the 6x3 factorisation is ~150 flops and is
// recomputed by every observation of the point rather than exchanged); the first observation of a point also
// writes the point's (tri, t, dinv).
template <typename T>
__global__ __launch_bounds__(256) void k_more_trial(int K, int Ml, const int *__restrict__ obs_pt, const int *__restrict__ pt_ptr,
                                                    const T *__restrict__ lam, const T *__restrict__ rec0, const T *__restrict__ tri0,
                                                    const T *__restrict__ tvec0, T *__restrict__ rec, T *__restrict__ dinv,
                                                    T *__restrict__ tvec, T *__restrict__ tri,
                                                    T *__restrict__ mQl = nullptr /* [Ml][9]: the lambda rows of the block's thin Q */,
                                                    T *__restrict__ mQR = nullptr /* [Ml][9]: its R1 rows */)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const int j = obs_pt[i];
    const T sl = tsqrt(*lam);
    const T r00 = tri0[j], r01 = tri0[(size_t)Ml + j], r02 = tri0[2 * (size_t)Ml + j], r11 = tri0[3 * (size_t)Ml + j],
            r12 = tri0[4 * (size_t)Ml + j], r22 = tri0[5 * (size_t)Ml + j];
    T W[3][3] = {{r00, 0, 0}, {r01, r11, 0}, {r02, r12, r22}}; // W[c][r]: column c, R1 rows
    T Rt[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, tau[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const T xn = W[c][0] * W[c][0] + W[c][1] * W[c][1] + W[c][2] * W[c][2];
        const T beta = -tsqrt(sl * sl + xn); // pivot = lambda row c (untouched by the earlier reflectors), lambda > 0
        tau[c] = (beta - sl) / beta;
        const T sc = (T)1.0 / (sl - beta);
#pragma unroll
        for (int r = 0; r < 3; r++) W[c][r] *= sc;
        Rt[c][c] = beta;
#pragma unroll
        for (int c2 = c + 1; c2 < 3; c2++) {
            const T w = (W[c][0] * W[c2][0] + W[c][1] * W[c2][1] + W[c][2] * W[c2][2]) * tau[c];
            Rt[c][c2] = -w;
#pragma unroll
            for (int r = 0; r < 3; r++) W[c2][r] -= W[c][r] * w;
        }
    }
    T Ql[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, QR[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int h = 2; h >= 0; h--)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const T w = (Ql[h][c] + W[h][0] * QR[0][c] + W[h][1] * QR[1][c] + W[h][2] * QR[2][c]) * tau[h];
            Ql[h][c] -= w;
#pragma unroll
            for (int r = 0; r < 3; r++) QR[r][c] -= W[h][r] * w;
        }
    const T m0 = tvec0[j], m1 = tvec0[(size_t)Ml + j], m2 = tvec0[2 * (size_t)Ml + j];
    T t[3];
#pragma unroll
    for (int c = 0; c < 3; c++) t[c] = QR[0][c] * m0 + QR[1][c] * m1 + QR[2][c] * m2;
    if (i == pt_ptr[j]) {
#pragma unroll
        for (int c = 0; c < 3; c++) { dinv[(size_t)c * Ml + j] = 1; tvec[(size_t)c * Ml + j] = t[c]; }
        tri[j] = Rt[0][0]; tri[(size_t)Ml + j] = Rt[0][1]; tri[2 * (size_t)Ml + j] = Rt[0][2];
        tri[3 * (size_t)Ml + j] = Rt[1][1]; tri[4 * (size_t)Ml + j] = Rt[1][2]; tri[5 * (size_t)Ml + j] = Rt[2][2];
        if (mQl) {
#pragma unroll
            for (int h = 0; h < 3; h++)
#pragma unroll
                for (int c = 0; c < 3; c++) { mQl[9 * (size_t)j + 3 * h + c] = Ql[h][c]; mQR[9 * (size_t)j + 3 * h + c] = QR[h][c]; }
        }
    }
    const T *z = rec0 + (size_t)i * BA_REC;
    T *o = rec + (size_t)i * BA_REC;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const T a0 = z[3 * c], a1 = z[3 * c + 1], a2 = z[3 * c + 2];
        const T z0 = a0 * QR[0][0] + a1 * QR[1][0] + a2 * QR[2][0], z1 = a0 * QR[0][1] + a1 * QR[1][1] + a2 * QR[2][1],
                z2 = a0 * QR[0][2] + a1 * QR[1][2] + a2 * QR[2][2];
        o[3 * c] = z0; o[3 * c + 1] = z1; o[3 * c + 2] = z2;
    }
    o[BA_REC_DINV] = 1; o[BA_REC_DINV + 1] = 1; o[BA_REC_DINV + 2] = 1;
}

// MOREQR, QR only (round 4; BacktrackLevMarqMore.h:297-345): what the per-point QRs of [sqrt(lambda) I3 ; R1_j] leave for the camera
// columns, written densely for the Householder QR of ba_qr.hip.h -- the inner counterpart of J2bot (k_qrkit_build):
//   rows 6 j .. 6 j + 5 (point j): (I - Q Q^T) [0 ; R12_j] with the block's thin Q = [Ql ; QR] (6 x 3), i.e. for the observation ia of
//     camera a:  lambda rows  -Ql Z_ia^T,  R1 rows  Z0_ia^T - QR Z_ia^T   (Z0 = R12^T of the outer factorisation, Z = Z0 QR: both 9 x 3 records);
//     right-hand side column D: (I - Q Q^T) [0 ; -q1_j] = [-Ql t ; (-q1) - QR t],  t = QR^T (-q1)  (tvec0 = -q1, tvec = t);
//   rows 6 M ...: R22 (D x D, upper triangle) and the head of Q^T (-qtb2) from the outer dense QR, then sqrt(lambda) I_D (k_more_tail).
// One thread per observation; the first observation of a point also writes its right-hand side rows.
template <typename T>
__global__ __launch_bounds__(256) void k_more_build(int K, int Ml, int D, const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                                                    const int *__restrict__ pt_ptr, const T *__restrict__ rec0, const T *__restrict__ rec,
                                                    const T *__restrict__ mQl, const T *__restrict__ mQR, const T *__restrict__ tvec0,
                                                    const T *__restrict__ tvec, T *__restrict__ A, size_t lda)
{
    const int ia = blockIdx.x * 256 + threadIdx.x;
    if (ia >= K) return;
    const int j = obs_pt[ia], a = obs_cam[ia], b = pt_ptr[j], e = pt_ptr[j + 1];
    for (int i2 = b; i2 < ia; i2++) // (several observations of one camera by one point: the first writes the sum of their blocks -- k_qrkit_build)
        if (obs_cam[i2] == a) return;
    const size_t r0 = 6 * (size_t)j;
    T Ql[9], QR[9];
#pragma unroll
    for (int q = 0; q < 9; q++) { Ql[q] = mQl[9 * (size_t)j + q]; QR[q] = mQR[9 * (size_t)j + q]; }
    T z0[27], z[27];
#pragma unroll
    for (int q = 0; q < 27; q++) { z0[q] = rec0[(size_t)ia * BA_REC + q]; z[q] = rec[(size_t)ia * BA_REC + q]; }
    for (int i2 = ia + 1; i2 < e; i2++)
        if (obs_cam[i2] == a) {
#pragma unroll
            for (int q = 0; q < 27; q++) { z0[q] += rec0[(size_t)i2 * BA_REC + q]; z[q] += rec[(size_t)i2 * BA_REC + q]; }
        }
    T *colbase = A + (size_t)(9 * a) * lda + r0;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const T z_0 = z[3 * c], z_1 = z[3 * c + 1], z_2 = z[3 * c + 2];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            colbase[(size_t)c * lda + rr] = -(Ql[3 * rr] * z_0 + Ql[3 * rr + 1] * z_1 + Ql[3 * rr + 2] * z_2);
            colbase[(size_t)c * lda + 3 + rr] = z0[3 * c + rr] - (QR[3 * rr] * z_0 + QR[3 * rr + 1] * z_1 + QR[3 * rr + 2] * z_2);
        }
    }
    if (ia == pt_ptr[j]) {
        const size_t M = (size_t)Ml;
        const T m0 = tvec0[j], m1 = tvec0[M + j], m2 = tvec0[2 * M + j], t0 = tvec[j], t1 = tvec[M + j], t2 = tvec[2 * M + j];
        const T mq[3] = {m0, m1, m2};
        T *rhs = A + (size_t)D * lda + r0;
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            rhs[rr] = -(Ql[3 * rr] * t0 + Ql[3 * rr + 1] * t1 + Ql[3 * rr + 2] * t2);
            rhs[3 + rr] = mq[rr] - (QR[3 * rr] * t0 + QR[3 * rr + 1] * t1 + QR[3 * rr + 2] * t2);
        }
    }
}
// the tail of that matrix (shard 0 only): R22 | c2 from the outer factorisation (R22buf: D x (D + 1), column-major ld D), sqrt(lambda) I_D
template <typename T>
__global__ __launch_bounds__(256) void k_more_tail(int Ml, int D, const T *__restrict__ R22buf, const T *__restrict__ lam, T *__restrict__ A, size_t lda, T dbg_scale = (T)1)
{
    const int c = blockIdx.x; // column 0 .. D (D = the right-hand side)
    size_t rR = 6 * (size_t)Ml, rL = rR + (size_t)D;
    if (dbg_scale < (T)0) { dbg_scale = -dbg_scale; rL = rR; rR = rL + (size_t)D; } // (experiment: the sqrt(lambda) rows in front of R22)
    const int top = c < D ? c : D - 1;
    for (int i = threadIdx.x; i <= top; i += 256) A[(size_t)c * lda + rR + i] = R22buf[(size_t)c * D + i];
    if (c < D && threadIdx.x == 0) A[(size_t)c * lda + rL + c] = tsqrt(*lam) * dbg_scale;
}
// R22 and the head of the transformed right-hand side out of a factored matrix (its first D rows), conditional on the step control
template <typename T>
__global__ __launch_bounds__(256) void k_copy_r22(int D, const T *__restrict__ A, size_t lda, T *__restrict__ R22buf, const int *__restrict__ go)
{
    if (go && *go == 0) return;
    const int c = blockIdx.x;
    const int top = c < D ? c : D - 1;
    for (int i = threadIdx.x; i < D; i += 256) R22buf[(size_t)c * D + i] = i <= top ? A[(size_t)c * lda + i] : (T)0;
}

// diagnostic (BA_DBG_ATB): out[c] = sum_r A[r][c] A[r][D] over the first `rows` rows -- A^T b of a built matrix, before it is factored
template <typename T>
__global__ __launch_bounds__(256) void k_dbg_atb(int rows, int D, const T *__restrict__ A, size_t lda, T *__restrict__ out)
{
    __shared__ T red[4];
    const int c = blockIdx.x;
    T a = 0;
    for (int r = threadIdx.x; r < rows; r += 256) a += A[(size_t)c * lda + r] * A[(size_t)D * lda + r];
    a = block_reduce<T, false>(a, red);
    if (threadIdx.x == 0) out[c] = a;
}

// diagnostic (BA_DBG_QRCHECK): the self-check of a least-squares solve  min || A y - b ||  on a copy of the matrix kept from before its
// factorisation: r = b - A y (thread per row), then out[c] = A(:, c)^T r and out[D + c] = A(:, c)^T b  (k_dbg_atb's layout, two passes)
template <typename T>
__global__ __launch_bounds__(256) void k_dbg_resid(int rows, int D, const T *__restrict__ A, size_t lda, const T *__restrict__ y, T *__restrict__ r)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    T a = A[(size_t)D * lda + i];
    for (int c = 0; c < D; c++) a -= A[(size_t)c * lda + i] * y[c];
    r[i] = a;
}
template <typename T>
__global__ __launch_bounds__(256) void k_dbg_atv(int rows, int D, const T *__restrict__ A, size_t lda, const T *__restrict__ v, T *__restrict__ out)
{
    __shared__ T red[4];
    const int c = blockIdx.x;
    T a = 0;
    for (int r = threadIdx.x; r < rows; r += 256) a += A[(size_t)c * lda + r] * v[r];
    a = block_reduce<T, false>(a, red);
    if (threadIdx.x == 0) out[c] = a;
}

// ---- K5: Schur complement / reduced camera matrix, pair-owner form -------------------------------------------
// S_ab = delta_ab (lambda I + sum A^T A) - sum_j Z_a diag(dinv_j) Z_b^T  (src/Eigen_ext/BacktrackLevMarqQRChol.h:334-341:
// J2bot^T J2bot; src/Eigen_ext/BacktrackLevMarqCholesky.h:274-278: the Schur complement inside the LDL^T).
//
// For one camera pair (a, b) the sum over the points both cameras see is a contraction over (point, coordinate):
//   [Z_a(e0) D(e0) | Z_a(e1) D(e1) | ...] (9 x 3n)  times  [Z_b(e0) | Z_b(e1) | ...]^T (3n x 9)
// -- the one place of the assembly where the matrix cores fit: v_mfma_{f64,f32}_16x16x4 with the 9 x 9 block in the corner of
// the 16 x 16 tile, four k-steps = 4/3 of an entry per instruction.  What it buys is not flops (a third of the tile is used) but
// LOADS: lane (i, k) fetches exactly the A element (i, k) and the B element (k, i) it feeds to the instruction, so every Z
// element is read once per entry (54 + 6 scalars) -- the lane-per-output form of round 1 read each of them 3 or 9 times
// (405 loads per entry, two entries in flight, 122 us at config 4).  Column 9 of B carries t of the point for self entries (row
// observation == column observation, diagonal pairs; the entry list holds ~point in place of the column observation), so column
// 9 of the tile is sum Z (dinv o t), the reduced-rhs term, for free.
//
// The kernel is a gather of 2 records (256 B each in fp64) per entry, and what bounds it is neither bytes nor flops but
// INSTRUCTION ISSUE (rocprofv3 counters of the first MFMA version: the SIMDs issued in 72 % of the kernel's cycles, 39 vector
// instructions per MFMA -- 64-bit address arithmetic, masks, index shuffles -- while doubling the L2 hit rate or the occupancy
// changed nothing).  So a lane owns one ENTRY of a group of four (k = entry, the three MFMAs of the group run over the point's
// three coordinates): one index shuffle pair and one 32-bit offset per three MFMAs, raw-buffer loads with immediate offsets (no
// 64-bit address arithmetic at all), and masking by out-of-range offsets (a raw buffer returns 0 there): lanes 9..15 of a row of
// 16 and entries past the end of a chunk cost no select.  The loads of the next eight entries are in flight while the current
// eight are multiplied.  (Also measured and dropped: staging whole records through LDS with 16-byte loads -- fewer, wider
// requests, but a longer dependent chain per batch: 134 against 100 us.)
//
// One wavefront per chunk of <= 64 entries of one pair, persistent: the host deals the chunks to the wavefronts of the grid so that
// every wavefront gets the same amount of work (longest-processing-time-first over the batch counts: chunk sizes run from 1 to 64
// entries, and a plain stride gave the busiest wavefront twice the mean -- the kernel then ran 116 us with wavefronts alive 47 us on
// average).  A wavefront walks its own list (wave_ptr) and has the descriptor of the chunk after next and the entry indices of the
// next chunk in flight while it works on the current one (one int4 per chunk: first entry, count | single-chunk flag, cameras
// hi | lo << 16, chunk id; one int2 per entry).  Optionally (nband = 8) the chunk list, which is sorted by (row camera, column
// camera), is first cut into one range per XCD (workgroups with equal blockIdx % 8 share an XCD under the observed round-robin
// placement), so that the records of neighbouring row cameras meet in one L2: L2 hits 46 % instead of 31 %.  A pair with a single chunk (almost all
// off-diagonal pairs) writes its block of S directly; pairs with several chunks (diagonal pairs, heavy pairs) leave partial
// tiles in a slab that k_schur_reduce sums in chunk order.  No atomics; the summation order is fixed by the static entry order,
// so the result is run-to-run reproducible.
#define BA_CHUNK_SINGLE (1 << 16) /* flag in chunk_info.y: the only chunk of its pair */
typedef int ba_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ba_bufload(__amdgpu_buffer_rsrc_t r, unsigned off, const double *)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ float ba_bufload(__amdgpu_buffer_rsrc_t r, unsigned off, const float *)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
// three consecutive scalars (a lane's row of Z, a point's 1 / D) with as few requests as the ISA allows: 16 + 8 bytes (fp64), 12 (fp32)
typedef int ba_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ba_bufload3(__amdgpu_buffer_rsrc_t r, unsigned off, double &x0, double &x1, double &x2)
{
    const ba_v4i lo = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    const ba_v2i hi = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(off + 16), 0, 0);
    x0 = __builtin_bit_cast(double, ba_v2i{lo.x, lo.y});
    x1 = __builtin_bit_cast(double, ba_v2i{lo.z, lo.w});
    x2 = __builtin_bit_cast(double, hi);
}
__device__ __forceinline__ void ba_bufload3(__amdgpu_buffer_rsrc_t r, unsigned off, float &x0, float &x1, float &x2)
{
    // (three 4-byte requests inside ONE 128-byte line.  A 12-byte load gave wrong sums here in round 3; round 4 found why
    // (scripts/experiments/buf96_trigger.hip): ROCm 7.2's clang narrows __builtin_amdgcn_raw_buffer_load_b96 to ONE buffer_load_dword
    // and copies that dword into all three results when the elements are bit-cast to float one by one -- a compiler defect, not the
    // instruction or the descriptor.  The fp32 record is a single cache line: there is no second line request to save.)
    x0 = ba_bufload(r, off, (const float *)nullptr);
    x1 = ba_bufload(r, off + 4, (const float *)nullptr);
    x2 = ba_bufload(r, off + 8, (const float *)nullptr);
}
// KO (knock-out experiment only, BA_SCHUR_KNOCKOUT=1|2: wrong results, the time tells): 1 = every ROW record is read from the first 1024
// records (always cache-resident), 2 = every COLUMN record, 3 = both -- what an order of the work that read that side once per camera could save at most.
// GB: groups of four entries per batch = what one memory round trip brings in per wavefront (two batches in flight).  The kernel is
// LATENCY-bound, not traffic-bound -- round 4's knock-outs at config 5: every record gather served from cache, 2.19 -> 1.71 ms -- so
// where the grid runs two workgroups per CU (records beyond the Infinity Cache: 256 registers per wavefront are there) the batches
// are twice or four times as deep.
template <typename T, bool SCALED /* dinv != 1: CHOLESKY */, int KO = 0, int GB = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GB == 2 ? 4 : 1))) void k_schur_pairs(const int *__restrict__ wave_ptr, int nband, const int4 *__restrict__ chunk_info,
                                                     const int2 *__restrict__ ent, const T *__restrict__ rec, unsigned rec_bytes,
                                                     const T *__restrict__ tvec, int Ml, T *__restrict__ slab, const T *__restrict__ V,
                                                     const T *__restrict__ gc, int D, int ld, T *__restrict__ S)
{
    const int lane = threadIdx.x & 63;
    // wavefront index: the workgroups with equal blockIdx % nband (one XCD, observed) own one band of the chunk list
    const int wq = ((blockIdx.x % nband) * (gridDim.x / nband) + blockIdx.x / nband) * 4 + (threadIdx.x >> 6);
    int slot = wave_ptr[wq];
    const int slot1 = wave_ptr[wq + 1];
    if (slot >= slot1) return; // (a whole wavefront leaves; the kernel has no workgroup barrier)
    // the records as a raw buffer: 32-bit byte offsets, immediate offsets in the instruction, 0 for every offset >= rec_bytes
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(rec), 0, (int)rec_bytes, 0x00020000);
    constexpr unsigned SZ = sizeof(T), RB = BA_REC * SZ, OOB = 0xfffffff0u - 64 * SZ;
    const int i = lane & 15, k = lane >> 4;
    const bool la = i < 9;
    const unsigned lane_off = 3 * i * SZ; // element (i, 0) of Z
    typedef typename ba_acc<T>::type acc_t;
    auto entry_of = [&](const int4 ci) { // this lane's entry of the chunk (lanes past the end shadow the last entry, their products are masked)
        const int n = ci.y & 0xffff;
        return ent[ci.x + (lane < n ? lane : n - 1)];
    };
    int4 ci0 = chunk_info[slot];
    int4 ci1 = chunk_info[min(slot + 1, slot1 - 1)];
    int2 en0 = entry_of(ci0);
    for (; slot < slot1; slot++) {
        const int2 en1 = entry_of(ci1);                          // next chunk's indices: arrive under this chunk's work
        const int4 ci2 = chunk_info[min(slot + 2, slot1 - 1)];   // descriptor of the chunk after next
        const int n = ci0.y & 0xffff;
        const int hi = ci0.z & 0xffff, lo = (unsigned)ci0.z >> 16, g = ci0.w;
        const bool diag = hi == lo; // (uniform) only diagonal pairs have self entries, i.e. a reduced-rhs column
        const int ia_l = en0.x, ib_l = en0.y;
        acc_t acc;
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = 0;
        // (GB groups of four entries per batch: template parameter)
        struct batch_t { T a[3 * GB], d[3 * GB], b[3 * GB]; };
        auto fetch = [&](int t0, batch_t &o) { // operands of the entries 4 t0 .. 4 (t0 + GB) - 1: lane (i, k) takes entry k of each group
#pragma unroll
            for (int u = 0; u < GB; u++) {
                const int e = 4 * (t0 + u) + k;
                const bool ok = e < n;
                const int es = ok ? e : n - 1;
                const int ia = __shfl(ia_l, es, 64), ibr = __shfl(ib_l, es, 64);
                const bool self = ibr < 0; // self entry: the column observation is the row observation, ibr = ~point
                const unsigned ra = (unsigned)((KO & 1) ? (ia & 1023) : ia) * RB, rb = (unsigned)((KO & 2) ? ((self ? ia : ibr) & 1023) : (self ? ia : ibr)) * RB;
                const unsigned oa = (ok && la) ? ra + lane_off : OOB, ob = (ok && la) ? rb + lane_off : OOB;
                const unsigned od = (ok && la) ? ra + BA_REC_DINV * SZ : OOB;
#ifdef BA_SCHUR_NARROW_LOADS
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    o.a[3 * u + m] = ba_bufload(rsrc, oa + m * SZ, (const T *)nullptr);
                    if (SCALED) o.d[3 * u + m] = ba_bufload(rsrc, od + m * SZ, (const T *)nullptr);
                    o.b[3 * u + m] = ba_bufload(rsrc, ob + m * SZ, (const T *)nullptr);
                }
#else
                ba_bufload3(rsrc, oa, o.a[3 * u], o.a[3 * u + 1], o.a[3 * u + 2]);
                if (SCALED) ba_bufload3(rsrc, od, o.d[3 * u], o.d[3 * u + 1], o.d[3 * u + 2]);
                ba_bufload3(rsrc, ob, o.b[3 * u], o.b[3 * u + 1], o.b[3 * u + 2]);
#endif
                if (diag && i == 9 && ok && self) { // column 9 of B: t of the point (reduced rhs)
#pragma unroll
                    for (int m = 0; m < 3; m++) o.b[3 * u + m] = tvec[(size_t)m * Ml + (~ibr)];
                }
            }
        };
        auto multiply = [&](const batch_t &o) {
#pragma unroll
            for (int q = 0; q < 3 * GB; q++) acc = ba_mfma(SCALED ? o.a[q] * o.d[q] : o.a[q], o.b[q], acc);
        };
        // two register sets, ping-pong (no copies): the next batch is in flight under the MFMAs of the current one
        const int ngroups = (n + 3) >> 2;
        batch_t b0, b1;
        fetch(0, b0);
        for (int t0 = 0;;) { // (uniform branches)
            if (t0 + GB < ngroups) fetch(t0 + GB, b1);
            multiply(b0);
            t0 += GB;
            if (t0 >= ngroups) break;
            if (t0 + GB < ngroups) fetch(t0 + GB, b0);
            multiply(b1);
            t0 += GB;
            if (t0 >= ngroups) break;
        }
        // tile element (row r, column j = i): r = ba_crow(k, v)
        if (ci0.y & BA_CHUNK_SINGLE) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int r = ba_crow<T>(k, v);
                if (r >= 9) continue;
                if (i < 9) {
                    T val = -acc[v];
                    if (diag) val += V[(size_t)hi * 81 + 9 * r + i];
                    S[(size_t)(9 * lo + i) * ld + 9 * hi + r] = val;
                } else if (i == 9 && diag) {
                    const T gg = gc[9 * hi + r];
                    S[(size_t)(9 * hi + r) * ld + D] = gg - acc[v];
                    S[(size_t)(9 * hi + r) * ld + D + 1] = gg;
                }
            }
        } else {
            T *o = slab + (size_t)g * BA_SLAB;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int r = ba_crow<T>(k, v);
                if (r >= 9) continue;
                if (i < 9) o[9 * r + i] = acc[v];
                else if (i == 9) o[81 + r] = acc[v];
            }
        }
        ci0 = ci1; ci1 = ci2; en0 = en1;
    }
}

// Sum the chunk partials of each pair with several chunks in chunk order and write the 9x9 block of S (lower block triangle;
// diagonal blocks in full); pairs without any entry get their zero (or J_c^T J_c) block here too.  Row D of S receives the reduced rhs, row D+1 the camera gradient g_c (both travel through the
// same all-reduce as S when the problem is sharded).  lambda I is added later by k_post_reduce (once, after the sum
// over shards).
// Single shard (lam != nullptr): there is no sum over shards to wait for, so k_post_reduce's work rides on this launch -- lambda
// goes onto the diagonal where the diagonal blocks are written (always here: the host never marks a diagonal pair BA_CHUNK_SINGLE;
// same order of additions as k_post_reduce: (V - s) + lambda), row D + 1 is not written at all, and `post_blocks` more workgroups
// at the end of the grid copy g_c out, clear the rows below the rhs row, give the padding its unit diagonal and arm the backward
// sweep's vectors.
// (Eight lanes per output, their partial sums meeting in a butterfly, were measured here and in k_cam_gram_reduce: config 2 7250 ->
// 7740 LM it/s -- but another order of additions, and at lambda <= 1e-9 (cond 1e19 and worse) that is enough to move single trials
// of the referee tests across their bounds.  The order of the additions stays; only more of their loads are in flight.)
template <typename T>
__global__ __launch_bounds__(192) void k_schur_reduce(int nred, const int *__restrict__ red_pairs, int D, int ld, const int *__restrict__ pair_hi,
                                                      const int *__restrict__ pair_lo, const int *__restrict__ pair_chunk_ptr,
                                                      const T *__restrict__ slab, const T *__restrict__ V,
                                                      const T *__restrict__ gc, T *__restrict__ S, const T *__restrict__ lam = nullptr,
                                                      int post_blocks = 0, int Dp = 0, T *__restrict__ gc_out = nullptr, T *__restrict__ xarm = nullptr)
{
    if (post_blocks && blockIdx.x >= gridDim.x - post_blocks) { // one wave per column (k_post_reduce's layout)
        const int c = (blockIdx.x - (gridDim.x - post_blocks)) * 3 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (c >= Dp) return;
        T *col = S + (size_t)c * ld;
        if (lane == 0) {
            xarm[c] = ba_sentinel<T>();
            xarm[Dp + c] = ba_sentinel<T>();
            if (c < D) gc_out[c] = gc[c];
        }
        if (c < D) {
            for (int rr = D + 1 + lane; rr < Dp; rr += 64) col[rr] = 0;
        } else {
            for (int rr = c + lane; rr < Dp; rr += 64) col[rr] = (rr == c) ? (T)1 : (T)0;
        }
        return;
    }
    const int idx = blockIdx.x * 192 + threadIdx.x;
    const int q = idx / BA_SLAB, e = idx - q * BA_SLAB;
    if (q >= nred || e >= 90) return;
    const int p = red_pairs[q]; // the pairs with no chunk or with several, and every diagonal pair (k_schur_pairs writes the other single-chunk ones itself)
    const int hi = pair_hi[p], lo = pair_lo[p];
    if (e >= 81 && hi != lo) return;
    T s4[4] = {0, 0, 0, 0}; // four interleaved partial sums (fixed order); sixteen loads in flight for the long lists (k_cam_gram_reduce)
    const int c1 = pair_chunk_ptr[p + 1];
    int c = pair_chunk_ptr[p];
    for (; c + 15 < c1; c += 16) {
        T x[16];
#pragma unroll
        for (int u = 0; u < 16; u++) x[u] = slab[(size_t)(c + u) * BA_SLAB + e];
#pragma unroll
        for (int u = 0; u < 16; u++) s4[u & 3] += x[u];
    }
    for (; c + 3 < c1; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) s4[u] += slab[(size_t)(c + u) * BA_SLAB + e];
    }
    for (; c < c1; c++) s4[0] += slab[(size_t)c * BA_SLAB + e];
    const T s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if (e < 81) {
        const int rr = e / 9, cc = e - 9 * rr;
        T v = -s;
        if (hi == lo) v += V[(size_t)hi * 81 + e];
        if (lam && hi == lo && rr == cc) v += *lam;
        S[(size_t)(9 * lo + cc) * ld + 9 * hi + rr] = v;
    } else {
        const int cc = e - 81;
        const T g = gc[9 * hi + cc];
        S[(size_t)(9 * hi + cc) * ld + D] = g - s;
        if (!lam) S[(size_t)(9 * hi + cc) * ld + D + 1] = g;
    }
}

// After the (optional) all-reduce: add lambda to the diagonal, copy the summed g_c out of row D+1, clear the
// augmented rows D+1.. and give the padding a unit diagonal so that the blocked LDL^T can run over whole tiles; arm the
// solution vector with the hand-off sentinel.
template <typename T>
__global__ __launch_bounds__(256) void k_post_reduce(int D, int Dp, int ld, const T *__restrict__ lam, T *__restrict__ S, T *__restrict__ gc_out,
                                                     T *__restrict__ xarm)
{
    // one wave per column, lane = row offset: the cleared rows of a column are contiguous (coalesced stores)
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= Dp) return;
    T *col = S + (size_t)c * ld;
    if (lane == 0) { // the solution vector of the backward sweep and its helpers' hand-over vector behind it (k_ldlt_backflow polls them entry by entry)
        xarm[c] = ba_sentinel<T>();
        xarm[Dp + c] = ba_sentinel<T>();
    }
    if (c < D) {
        if (lane == 0) {
            col[c] += *lam;
            gc_out[c] = col[D + 1]; // (this lane clears that entry below, in program order)
        }
        for (int rr = D + 1 + lane; rr < Dp; rr += 64) col[rr] = 0;
    } else {
        for (int rr = c + lane; rr < Dp; rr += 64) col[rr] = (rr == c) ? (T)1 : (T)0;
    }
}

// ---- multi-GPU: pack / unpack the block-lower trapezoid of S around the all-reduce ----------------------------------
// Only rows >= 64 (c / 64) of column c carry data (lower triangle + the augmented rows D, D+1 at the bottom); packing
// them into one contiguous buffer halves the bytes every rank sends over xGMI.  off[p] = start of block column p.
// The packed buffer ends with one scalar: this shard's part of the energy of the latest linearisation (scal[eloc]) on the way
// out, the energy summed over the shards (scal[etot]) on the way back -- the energy of an accepted step rides on the next
// trial's all-reduce instead of costing a collective of its own.
template <typename T, bool UNPACK>
__global__ __launch_bounds__(256) void k_pack_lower(int Dp, int ld, T *__restrict__ S, T *__restrict__ buf, size_t ntail, T *__restrict__ scal,
                                                    int eloc, int etot)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (UNPACK) scal[etot] = buf[ntail];
        else buf[ntail] = scal[eloc];
    }
    const int c = blockIdx.y;                 // column
    const int p = c >> 6, r0 = p << 6;        // block column, first kept row
    const int h = Dp - r0;                    // kept rows of this column
    // offset of block column p: sum_{q<p} 64 (Dp - 64 q) ; plus the columns before c inside the block
    const size_t off = (size_t)64 * ((size_t)p * Dp - (size_t)32 * p * (p - 1)) + (size_t)(c - r0) * h;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < h; r += gridDim.x * 256) {
        if (UNPACK) S[(size_t)c * ld + r0 + r] = buf[off + r];
        else buf[off + r] = S[(size_t)c * ld + r0 + r];
    }
}

// Distributed factor (BA_DIST_FACTOR): the same trapezoid packed OWNER BY OWNER for the reduce-scatter -- block column p belongs to
// rank p % world and sits at own_off[p] (in units of 64 scalars) in a buffer of `world` chunks of equal length.  Behind the chunks, at
// `small`: the camera gradient g_c (row D + 1 of the columns, D scalars) and this shard's energy -- needed on EVERY rank, they go
// through a small all-reduce of their own.  UNPACK: only this rank's block columns come back (the others are overwritten by the
// owners' broadcast panels as the factorisation reaches them); the summed g_c returns to row D + 1 of every column (k_post_reduce
// reads it there), the summed energy to scal[etot].
template <typename T, bool UNPACK>
__global__ __launch_bounds__(256) void k_pack_owner(int Dp, int D, int ld, T *__restrict__ S, T *__restrict__ buf, const int *__restrict__ own_off,
                                                    int world, int rank, size_t small, T *__restrict__ scal, int eloc, int etot)
{
    const int c = blockIdx.y;          // column
    const int p = c >> 6, r0 = p << 6; // block column, first kept row
    const int h = Dp - r0;             // kept rows of this column
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (c == 0) { if (UNPACK) scal[etot] = buf[small + D]; else buf[small + D] = scal[eloc]; }
        if (c < D) { if (UNPACK) S[(size_t)c * ld + D + 1] = buf[small + c]; else buf[small + c] = S[(size_t)c * ld + D + 1]; }
    }
    if (UNPACK && p % world != rank) return;
    const size_t off = (size_t)64 * (size_t)own_off[p] + (size_t)(c - r0) * h;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < h; r += gridDim.x * 256) {
        if (UNPACK) { if (r0 + r != D + 1 || c >= D) S[(size_t)c * ld + r0 + r] = buf[off + r]; } // (row D + 1: the all-reduced g_c, written above)
        else buf[off + r] = S[(size_t)c * ld + r0 + r];
    }
}

// One block column's factor pieces, contiguous for ONE broadcast: [L: h x 64 (rows p0 .. of the block column of S) | Y = L D: h x 64
// (the same rows of Wp) | W = L11^-1: 64 x 64].  grid (ceil(h / 256), 2 * 64 + 1): y < 64 a column of L, < 128 a column of Y, 128 = W.
template <typename T, int NB, bool UNSTAGE>
__global__ __launch_bounds__(256) void k_panel_stage(int h, int p0, int ld, T *__restrict__ S, T *__restrict__ Wp, T *__restrict__ Winv, T *__restrict__ stage)
{
    const int y = blockIdx.y;
    if (y == 2 * NB) {
        for (int q = blockIdx.x * 256 + threadIdx.x; q < NB * NB; q += gridDim.x * 256) {
            if (UNSTAGE) Winv[q] = stage[(size_t)2 * h * NB + q]; else stage[(size_t)2 * h * NB + q] = Winv[q];
        }
        return;
    }
    const int j = y < NB ? y : y - NB;
    T *src = (y < NB ? S + (size_t)(p0 + j) * ld : Wp + (size_t)j * ld) + p0;
    T *dst = stage + (size_t)(y < NB ? 0 : h * NB) + (size_t)j * h;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < h; r += gridDim.x * 256) {
        if (UNSTAGE) src[r] = dst[r]; else dst[r] = src[r];
    }
}

// ---- K7 + K8 (points): back-substitution, point retraction, rho terms ----------------------------------------
// dx_p = tri^-1 (dinv o (t - sum_i Z_i^T dx_c[cam_i]))  (src/Eigen_ext/BacktrackLevMarqQRChol.h:343-360);
// x_test = x + dx_p (src/Optimization/BAFunctor.h:335-338); partial sums of dx^T (lambda dx + JtRes) (:375) and |dx|^2.
struct ba_cam_retract_args { int N; const void *cam, *dxc, *gc; void *cam_test, *scal; int dst; };
template <typename T>
__device__ __forceinline__ void ba_retract_cams(int N, const T *__restrict__ cam, const T *__restrict__ dxc, const T *__restrict__ gc,
                                                const T *__restrict__ lam, T *__restrict__ cam_test, T *__restrict__ scal, int dst, T *red);

// cr.N > 0: one more block at the end of the grid retracts the cameras (K8, below: a launch of its own otherwise); npart = the
// number of point blocks = the row length of partial[2][npart].
template <typename T, int LPP>
__global__ __launch_bounds__(256) void k_backsub(int Ml, const int *__restrict__ pt_ptr, const int *__restrict__ obs_cam,
                                                 const T *__restrict__ rec, const T *__restrict__ dinv, const T *__restrict__ tvec,
                                                 const T *__restrict__ tri, const T *__restrict__ dxc, const T *__restrict__ gp,
                                                 const T *__restrict__ pts, const T *__restrict__ lam, T *__restrict__ dxp, T *__restrict__ pts_test,
                                                 T *__restrict__ partial /* [2][npart] */, const int *__restrict__ pperm /* column permutation of the point's 3x3 block */,
                                                 int npart, ba_cam_retract_args cr)
{
    // LPP lanes per point: each lane forms Z_i^T dx_c for its observations (i = g, g + LPP, ...), a butterfly sum over
    // the group gives the point's 3-vector, lane 0 of the group finishes the 3x3 triangular solve.
    __shared__ T red[4];
    if (cr.N > 0 && (int)blockIdx.x == npart) {
        ba_retract_cams<T>(cr.N, (const T *)cr.cam, (const T *)cr.dxc, (const T *)cr.gc, lam, (T *)cr.cam_test, (T *)cr.scal, cr.dst, red);
        return;
    }
    const int gid = (blockIdx.x * 256 + threadIdx.x) / LPP, lg = threadIdx.x % LPP;
    const int j = gid < Ml ? gid : Ml - 1;
    const int b = pt_ptr[j], e = pt_ptr[j + 1];
    T s0 = 0, s1 = 0, s2 = 0;
    for (int i = b + lg; i < e; i += LPP) {
        const T *Z = rec + (size_t)i * BA_REC;
        const T *dc = dxc + 9 * obs_cam[i];
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const T d = dc[c];
            s0 += Z[3 * c] * d; s1 += Z[3 * c + 1] * d; s2 += Z[3 * c + 2] * d;
        }
    }
    s0 = group_sum<T, LPP>(s0); s1 = group_sum<T, LPP>(s1); s2 = group_sum<T, LPP>(s2);
    T rho = 0, dn = 0;
    if (gid < Ml && lg == 0 && b == e) {
        // a point without observations: its block of J'J + lambda I is lambda I with a zero gradient -> no step
        // (the per-observation elimination kernel never visits it, so its factors are not even written)
        dxp[j] = 0; dxp[(size_t)Ml + j] = 0; dxp[2 * (size_t)Ml + j] = 0;
        pts_test[j] = pts[j]; pts_test[(size_t)Ml + j] = pts[(size_t)Ml + j]; pts_test[2 * (size_t)Ml + j] = pts[2 * (size_t)Ml + j];
    } else if (gid < Ml && lg == 0) {
        const T lambda = *lam;
        const T u0 = (tvec[j] - s0) * dinv[j], u1 = (tvec[(size_t)Ml + j] - s1) * dinv[(size_t)Ml + j],
                u2 = (tvec[2 * (size_t)Ml + j] - s2) * dinv[2 * (size_t)Ml + j];
        const T y2 = u2 / tri[5 * (size_t)Ml + j];
        const T y1 = (u1 - tri[4 * (size_t)Ml + j] * y2) / tri[3 * (size_t)Ml + j];
        const T y0 = (u0 - tri[(size_t)Ml + j] * y1 - tri[2 * (size_t)Ml + j] * y2) / tri[j];
        // m_dx = colsPermutation() * m_dx (BacktrackLevMarqQRChol.h:360): position c of the pivoted block is coordinate perm[c]
        const int pp = pperm[j], q0 = pp & 3, q1 = (pp >> 2) & 3;
        const T x0 = q0 == 0 ? y0 : q1 == 0 ? y1 : y2, x1 = q0 == 1 ? y0 : q1 == 1 ? y1 : y2, x2 = q0 == 2 ? y0 : q1 == 2 ? y1 : y2;
        dxp[j] = x0; dxp[(size_t)Ml + j] = x1; dxp[2 * (size_t)Ml + j] = x2;
        pts_test[j] = pts[j] + x0;
        pts_test[(size_t)Ml + j] = pts[(size_t)Ml + j] + x1;
        pts_test[2 * (size_t)Ml + j] = pts[2 * (size_t)Ml + j] + x2;
        rho = x0 * (lambda * x0 + gp[j]) + x1 * (lambda * x1 + gp[(size_t)Ml + j]) + x2 * (lambda * x2 + gp[2 * (size_t)Ml + j]);
        dn = x0 * x0 + x1 * x1 + x2 * x2;
    }
    rho = block_reduce<T, false>(rho, red);
    dn = block_reduce<T, false>(dn, red);
    if (threadIdx.x == 0) { partial[blockIdx.x] = rho; partial[npart + blockIdx.x] = dn; }
}

// ---- K8 (cameras): BAFunctor::update_params (src/Optimization/BAFunctor.h:311-332) -----------------------------
// T += dT; R <- Rodrigues(d omega) R (identity when |d omega| <= 1e-6, src/MathUtils.h:66-82); f, k1, k2 += .
// Single block (N <= a few thousand cameras); also the camera part of the rho / |dx|^2 sums -> scal[dst..dst+1].
template <typename T>
__device__ __forceinline__ void ba_retract_cams(int N, const T *__restrict__ cam, const T *__restrict__ dxc,
                                                const T *__restrict__ gc, const T *__restrict__ lam, T *__restrict__ cam_test,
                                                T *__restrict__ scal, int dst, T *red)
{
    const T lambda = *lam;
    T rho = 0, dn = 0;
    for (int a = threadIdx.x; a < N; a += 256) {
        T p[9];
#pragma unroll
        for (int c = 0; c < 9; c++) {
            p[c] = dxc[9 * a + c];
            rho += p[c] * (lambda * p[c] + gc[9 * a + c]);
            dn += p[c] * p[c];
        }
        T R0[9];
#pragma unroll
        for (int c = 0; c < 9; c++) R0[c] = cam[(size_t)c * N + a];
        const T th = tsqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
        T dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (th > (T)1e-6) {
            const T J[9] = {0, -p[5], p[4], p[5], 0, -p[3], -p[4], p[3], 0};
            const T c1 = tsin(th) / th, c2 = ((T)1.0 - tcos(th)) / (th * th);
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {
                    T j2 = 0;
#pragma unroll
                    for (int k = 0; k < 3; k++) j2 += J[3 * rr + k] * J[3 * k + cc];
                    dR[3 * rr + cc] = dR[3 * rr + cc] + c1 * J[3 * rr + cc] + c2 * j2;
                }
        }
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                T s = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) s += dR[3 * rr + k] * R0[3 * k + cc];
                cam_test[(size_t)(3 * rr + cc) * N + a] = s;
            }
#pragma unroll
        for (int c = 0; c < 3; c++) cam_test[(size_t)(9 + c) * N + a] = cam[(size_t)(9 + c) * N + a] + p[c];
        cam_test[(size_t)12 * N + a] = cam[(size_t)12 * N + a] + p[6];
        cam_test[(size_t)13 * N + a] = cam[(size_t)13 * N + a] + p[7];
        cam_test[(size_t)14 * N + a] = cam[(size_t)14 * N + a] + p[8];
    }
    rho = block_reduce<T, false>(rho, red);
    dn = block_reduce<T, false>(dn, red);
    if (threadIdx.x == 0) { scal[dst] = rho; scal[dst + 1] = dn; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_retract_cams(int N, const T *__restrict__ cam, const T *__restrict__ dxc,
                                                      const T *__restrict__ gc, const T *__restrict__ lam, T *__restrict__ cam_test,
                                                      T *__restrict__ scal, int dst)
{
    __shared__ T red[4];
    ba_retract_cams<T>(N, cam, dxc, gc, lam, cam_test, scal, dst, red);
}

// ---- a-8: step control on the device ---------------------------------------------------------------------------
// The accept / rho / lambda / flat-line logic of BacktrackLevMarqQRChol.h:374-428 (== BacktrackLevMarqCholesky.h:299-353) as a
// one-lane kernel behind every trial, so that a whole LM run -- rejected retries included -- is enqueued without the host ever
// waiting for a scalar: the trial kernels read lambda from scal[], k_commit (x = xTest, :428) and the linearisation kernels of
// the next outer iteration look at lm->go.  Every trial leaves one row of the reference's table (:84-93) in a ring in pinned
// host memory and then bumps `done` (system-scope release): the host prints / forwards rows as they appear and throttles its
// enqueueing on `done`, nothing else.
#define BA_LM_RING 64
#define BA_DEV_FAILED (-100) /* status: a kernel raised the device error word (scal[err_slot]) */
template <typename T> struct ba_lm_dev {
    T lambda, lambda_inc, energy, hist0, hist1;
    T lam_min, lam_max, tol_fun, inc_base;
    int iter, trials, fun_evals, status, stop;
    int go;       // the last trial was accepted and x = xTest is to happen: k_commit and the linearisation behind it run
    int fresh;    // the linearisation at x has been redone since `energy` was set: take it from scal[SC energy slot]
    int max_iter, max_fun_ev, max_trials, deverr;
    // per-trial device times without events between the launches (the whole iteration is ONE graph): wall_clock64() at the start of
    // k_lm_control = the end of the trial (t_ctl), and at the end of the control segment's last kernel (t_end, k_reduce_scalars)
    long long t_ctl, t_end;
    int timed;   // t_ctl / t_end describe the iteration just before this one (not the host-synchronous first linearisation)
    int prev_go; // ... and that iteration linearised
    int prev_code; // decision of the previous trial (accepted + 2 stop): what this shard put into the guard slot of the scalar all-reduce
    int rec_fresh; // the fused linearisation (k_eval<T, true, 2>) has left the records of the NEXT trial: its k_elim_chol returns at once
};
// trial_ticks: t_ctl - previous t_end (elimination ... test energy); ctl_ticks_prev: the control segment (control, x = xTest,
// linearisation) that preceded this trial; both < 0 when unknown
// stop: 0 = the run goes on behind this row, 1 = this row ended it (the host enqueues trial n only when row n - LM_DEPTH carries
// stop == 0: the number of trials -- and of collectives -- a rank enqueues depends on device data alone), 2 = ended by a device
// error: the row is not part of the table
struct ba_lm_row { double iter, accepted, f, rho, lambda, lambda_used, e_test, dx_norm, trial_ticks, ctl_ticks_prev, prev_go, stop; };
struct ba_lm_host { int done, stop, status, pad; ba_lm_row rows[BA_LM_RING]; };
// indices into scal[]; guard: the slot that rides on the scalar all-reduce with every shard's previous decision (world = summands)
struct ba_lm_slots { int energy, etest, rho_p, rho_c, dn_p, dn_c, lambda, err, guard, world; };

// jobs (njobs > 0, single shard): the second stage of the trial's scalar reductions runs at the head of this launch instead of in a
// k_reduce_scalars launch of its own (same 256-thread order, so the sums are the same bits); sharded runs keep that launch, the
// all-reduce of the step scalars sits between the two.
#define BA_LM_JOBS 3 /* sum jobs k_lm_control can take: 256 threads each */
template <typename T>
__global__ __launch_bounds__(256 * BA_LM_JOBS) void k_lm_control(T *__restrict__ scal, ba_lm_dev<T> *__restrict__ lm, ba_lm_host *__restrict__ host,
                                                                 ba_lm_slots sl, ba_red_jobs jobs, int njobs)
{
    if (njobs > 0) { // 256 threads per job, side by side, each group in the order of a k_reduce_scalars block (ba_reduce_job, sum)
        __shared__ T red[BA_LM_JOBS][4];
        const int q = threadIdx.x >> 8, t = threadIdx.x & 255;
        T a = 0;
        if (q < njobs) {
            const T *src = (const T *)jobs.j[q].src;
            for (int k = t; k < jobs.j[q].n; k += 256) a += src[k];
        }
        a = wave_reduce<T, false>(a);
        if ((t & 63) == 0) red[q][t >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int j = 0; j < njobs; j++) scal[jobs.j[j].dst] = ((red[j][0] + red[j][1]) + red[j][2]) + red[j][3];
    }
    if (threadIdx.x != 0) return; // (thread 0 wrote the sums itself: it reads its own stores below)
    const long long now = (long long)wall_clock64();
    ba_lm_dev<T> s = *lm;
    if (s.stop) { // a trial enqueued behind the end of the run: it changes nothing
        if (s.go) lm->go = 0;
        return;
    }
    if (s.fresh) { // m_functor(x, r) at the top of the outer iteration (:257): the accepted point re-evaluated
        s.energy = scal[sl.energy];
        s.fresh = 0;
        s.fun_evals++;
    }
    const T e_test = scal[sl.etest];
    const T rs = scal[sl.rho_p] + scal[sl.rho_c];
    const T dn = scal[sl.dn_p] + scal[sl.dn_c];
    s.fun_evals++;
    const int t = s.trials++;
    const T lam_used = s.lambda;
    ba_lm_row row;
    row.trial_ticks = s.timed ? (double)(now - s.t_end) : -1.0;
    row.ctl_ticks_prev = s.timed ? (double)(s.t_end - s.t_ctl) : -1.0;
    row.prev_go = s.prev_go;
    s.t_ctl = now;
    s.timed = 1;
    row.iter = s.iter; row.f = (double)s.energy; row.lambda_used = (double)lam_used; row.e_test = (double)e_test; row.dx_norm = sqrt((double)dn);
    int go = 0;
    bool failed = false;
    if (scal[sl.guard] != (T)(sl.world * s.prev_code)) { // sharded: the decisions of the previous trial, summed over the shards, are
        // not `world` times this shard's -- the shards have parted (they run the same control on the same all-reduced scalars, so
        // this cannot happen unless a transport hands different sums to different ranks); stop loudly instead of drifting apart
        s.deverr = BA_DEVERR_DIVERGED;
        failed = true;
    } else if (scal[sl.err] != (T)0) { // a hand-off wait ran out inside this trial (ba_dense.hip.h): the step is garbage, stop loudly
        // (sharded: the SUM of the shards' error words -- it rides on the scalar all-reduce, so every shard stops on the same trial;
        // BA_DEVERR_ROW_FLAG counts in the low ten bits, BA_DEVERR_SWEEP above them)
        s.deverr = scal[sl.err] >= (T)BA_DEVERR_SWEEP ? BA_DEVERR_SWEEP : BA_DEVERR_ROW_FLAG;
        failed = true;
    }
    scal[sl.err] = 0;
    if (failed) {
        s.status = BA_DEV_FAILED;
        s.stop = 1;
        row.accepted = 0; row.rho = 0; row.lambda = (double)s.lambda;
    } else if (e_test < s.energy) { // :374-394
        const T rho = (s.energy - e_test) / rs;
        const T tmv = (T)2.0 * rho - (T)1.0;
        const T mul = (T)1.0 - tmv * tmv * tmv;
        const T third = (T)1.0 / (T)3.0;
        s.lambda *= (mul > third ? mul : third);
        s.lambda = s.lambda > s.lam_min ? s.lambda : s.lam_min;
        row.accepted = 1; row.rho = (double)rho; row.lambda = (double)s.lambda;
        s.lambda_inc = s.inc_base;
        s.energy = e_test;
        if (s.iter % 2) s.hist1 = s.energy; else s.hist0 = s.energy;
        const T maxf = s.hist0 > s.hist1 ? s.hist0 : s.hist1;
        const T diff = s.energy > maxf ? s.energy - maxf : maxf - s.energy;
        if (s.iter > 2 && diff < s.tol_fun * s.energy) { // flat-line: Success, and the loop leaves BEFORE x = xTest (:419-428)
            s.status = 0;
            s.stop = 1;
        } else {
            go = 1; // x = xTest, then the top of the next outer iteration (:243-254)
            s.iter++;
            s.fresh = 1;
            if (s.iter > s.max_iter) { s.status = 3; s.stop = 1; }
            else if (s.fun_evals > s.max_fun_ev) { s.status = 2; s.stop = 1; }
        }
    } else { // :395-410
        row.accepted = 0; row.rho = 0; row.lambda = (double)s.lambda;
        if (s.lambda > s.lam_max) { s.status = 1; s.stop = 1; }
        else {
            s.lambda *= s.lambda_inc;
            s.lambda_inc = sizeof(T) == 8 ? (T)pow((double)s.lambda_inc, 1.5) : (T)powf((float)s.lambda_inc, 1.5f);
        }
    }
    if (!s.stop && s.max_trials > 0 && s.trials >= s.max_trials) { s.stop = 1; s.status = -1; } // the max_trials extension: Running
    s.go = go;
    s.prev_go = go;
    s.prev_code = (row.accepted != 0 ? 1 : 0) + 2 * (s.stop != 0 ? 1 : 0);
    scal[sl.guard] = (T)s.prev_code;
    scal[sl.lambda] = s.lambda;
    row.stop = failed ? 2.0 : (s.stop ? 1.0 : 0.0);
    s.rec_fresh = 0; // this trial's records are spent; the linearisation behind an accepted step (same control segment) may leave the next ones
    *lm = s;
    host->rows[t % BA_LM_RING] = row;
    host->stop = s.stop;
    host->status = s.status;
    __threadfence_system();
    __hip_atomic_store(&host->done, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Self-test only (ba_solver_selftest(3)): keeps the stream busy for `ticks` of the 100 MHz wall clock, then ends by itself.
__global__ void k_spin(long long ticks)
{
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// x = xTest (BacktrackLevMarqQRChol.h:428), when the control kernel said so (go == nullptr: unconditionally, ba_solver_accept).
template <typename T>
__global__ __launch_bounds__(256) void k_commit(int ncam, int npts, const T *__restrict__ cam_test, const T *__restrict__ pts_test,
                                                T *__restrict__ cam, T *__restrict__ pts, const int *__restrict__ go)
{
    if (go && *go == 0) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < ncam) cam[idx] = cam_test[idx];
    else if (idx - ncam < npts) pts[idx - ncam] = pts_test[idx - ncam];
}

#endif
