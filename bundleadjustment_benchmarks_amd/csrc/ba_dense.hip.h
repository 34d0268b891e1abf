// ba_dense.hip.h -- K6: dense LDL^T of the reduced camera matrix and the triangular solves, gfx950.
//
// Stands in for Eigen::SimplicialLDLT on J2bot^T J2bot (src/Optimization/BAFunctor.h:106,
// src/Eigen_ext/BacktrackLevMarqQRChol.h:339-341) and, for the CHOLESKY symbol, for the camera part of the
// LDL^T of the whole J^T J + lambda I (src/Eigen_ext/BacktrackLevMarqCholesky.h:156,274-282).  The reduced matrix
// is 85-100 % block-dense, so it is factored as a dense matrix: right-looking, 64-wide block columns.
//
//   k_ldlt_panel   every workgroup factors the 64x64 diagonal block (redundantly: it is the critical path and a
//                  broadcast would cost a kernel boundary).  The block lives in registers, 4x4 elements per thread in
//                  a 16x16-cyclic distribution, so the work per pivot stays balanced as the active part shrinks; one
//                  barrier per pivot; the same row operations applied to I give W = L11^-1.  The rows below then
//                  need Y = A21 L11^-T = A21 W^T, which is a GEMM: done on the matrix cores.
//   k_ldlt_update  trailing update S_ij -= (L D)_i L_j^T on the matrix cores (v_mfma_f64_16x16x4_f64 /
//                  v_mfma_f32_16x16x4_f32) -- the one true contraction of the LM trial.
//   k_ldlt_backstep backward sweep L^T x = z, one launch per block column, using the stored W (x_p = W_p^T z_p).
//
// No pivoting, no square roots: D keeps the sign of a pivot, like SimplicialLDLT.  The right-hand side rides along
// as the extra matrix row `zrow` = D: after the factorisation that row holds D^-1 L^-1 b.
//
// MFMA fragment maps (cdna_hip_programming.md s3): A[i][k]: lane l holds i = l & 15, k = l >> 4; B[k][j]: lane l holds
// k = l >> 4, j = l & 15; C/D: col = l & 15, row = (l >> 4) + 4 v for f64 and 4 (l >> 4) + v for f32.
#ifndef BA_DENSE_HIP_H
#define BA_DENSE_HIP_H

#include <hip/hip_runtime.h>

typedef double ba_d4 __attribute__((ext_vector_type(4)));
typedef float ba_f4 __attribute__((ext_vector_type(4)));

template <typename T> struct ba_acc;
template <> struct ba_acc<double> { typedef ba_d4 type; };
template <> struct ba_acc<float> { typedef ba_f4 type; };

__device__ __forceinline__ ba_d4 ba_mfma(double a, double b, ba_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ba_f4 ba_mfma(float a, float b, ba_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename T> __device__ __forceinline__ int ba_crow(int lk, int v) { return sizeof(T) == 8 ? lk + 4 * v : 4 * lk + v; }

// 1/d to (nearly) full precision: hardware estimate + two Newton steps (the division expansion would sit on the
// critical path of every pivot).
__device__ __forceinline__ double ba_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float ba_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return r;
}

// Pinned (asm volatile) forms of the reciprocal chain: hipcc otherwise sinks the chain below the rank-1 FMAs it is
// meant to overlap with (it is only consumed by the next loop iteration).  One instruction per statement; the s_nop
// covers the trans-op -> VALU read hazard the compiler cannot see inside an asm.
__device__ __forceinline__ double ba_rcp_est(double d) { double r; asm volatile("v_rcp_f64 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(d)); return r; }
__device__ __forceinline__ float ba_rcp_est(float d) { float r; asm volatile("v_rcp_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(d)); return r; }
// 1 - d r
__device__ __forceinline__ double ba_fnma1(double d, double r) { double e; const double one = 1.0; asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(e) : "v"(d), "v"(r), "v"(one)); return e; }
__device__ __forceinline__ float ba_fnma1(float d, float r) { float e; const float one = 1.0f; asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e) : "v"(d), "v"(r), "v"(one)); return e; }
__device__ __forceinline__ double ba_fma_(double a, double b, double c) { double r; asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float ba_fma_(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

#define BA_NB 64

// Diagnostic build only (-DBA_STAMP, scripts/bench_dense.hip): cycle stamps of the pivot loop's segments.
#ifdef BA_STAMP
__device__ long long ba_stamp_acc[8 * 8];
#define BA_STAMP_DECL unsigned long long st_t0 = 0, st_t1 = 0; long long st_acc[6] = {0, 0, 0, 0, 0, 0};
#define BA_STAMP_GET(v) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define BA_STAMP_SEG(i) { BA_STAMP_GET(st_t1); st_acc[i] += (long long)(st_t1 - st_t0); st_t0 = st_t1; }
#else
#define BA_STAMP_DECL
#define BA_STAMP_GET(v)
#define BA_STAMP_SEG(i)
#endif

// Panel step for block column p0: rows [p0, nrows), pivots [p0, min(p0 + 64, ncols)).
//   S    : in place; on exit the block column holds L (strictly lower) and D (diagonal)
//   Wp   : ld x 64, column-major: Y = L D for the rows below the diagonal block (A operand of the trailing update)
//   Winv : 64 x 64 row-major: W = L11^-1 of this block (for the backward sweep)
// Grid: one workgroup per 64 rows below the diagonal block (at least one).
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_panel(int nrows, int ncols, int ld, int p0, T *__restrict__ S, T *__restrict__ Wp,
                                                    T *__restrict__ Winv)
{
    constexpr int R = NB / 16; // register blocks per dimension (16x16-cyclic ownership)
    __shared__ T colb[2][NB], roww[2][NB];
    __shared__ T Wl[NB][NB + 1];
    __shared__ T dinv[NB];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int nb = min(NB, ncols - p0);
    // 16x16-cyclic ownership: element (i, j) = (ty + 16 a, tx + 16 b)
    T a_[R][R], w_[R][R];
#pragma unroll
    for (int a = 0; a < R; a++)
#pragma unroll
        for (int b = 0; b < R; b++) {
            const int i = ty + 16 * a, j = tx + 16 * b;
            const int lo = min(i, j), hi = max(i, j);
            a_[a][b] = S[(size_t)(p0 + lo) * ld + p0 + hi]; // mirror the lower triangle
            w_[a][b] = (i == j) ? (T)1 : (T)0;
        }
    // Software-pipelined pivots.  A dependent f64 FMA costs ~32 cycles on gfx950, so the critical chain of one pivot
    //   d_k -> 1/d_k (estimate + 2 Newton steps) -> l_ik -> column k+1 -> LDS -> barrier -> LDS -> d_k+1
    // is pure latency.  The 32 rank-1 FMAs per thread (A and W) of pivot k are issued AFTER the barrier of pivot k and
    // interleaved by hand with the LDS reads and the reciprocal chain of pivot k+1 (the compiler would otherwise
    // serialise the two), so they fill the stall slots of that chain.  One barrier per pivot, double-buffered LDS,
    // two register sets for the multipliers (ping-pong, no copies).
    BA_STAMP_DECL
    struct PV { T l[R], y[R], wk[R], inv; };
    PV pv0, pv1;
    auto fetch0 = [&](PV &n) {
        const T d = colb[0][0];
#pragma unroll
        for (int a = 0; a < R; a++) n.l[a] = colb[0][ty + 16 * a];
#pragma unroll
        for (int b = 0; b < R; b++) { n.y[b] = colb[0][tx + 16 * b]; n.wk[b] = roww[0][tx + 16 * b]; }
        n.inv = ba_rcp(d);
#pragma unroll
        for (int a = 0; a < R; a++) n.l[a] = (ty + 16 * a > 0) ? n.l[a] * n.inv : (T)0;
#pragma unroll
        for (int b = 0; b < R; b++) n.y[b] = (tx + 16 * b > 0) ? n.y[b] : (T)0;
    };
    auto pivot = [&](const int kb, const int km, const PV &c, PV &n) {
        const int k = 16 * kb + km, buf = k & 1, kn = k + 1, kmn = kn & 15;
        // 1. finalise and publish column k+1 of A and row k+1 of W (temporaries; step 3 recomputes the same values)
        BA_STAMP_GET(st_t0);
        if (kn < nb) {
            const int kq = (km < 15) ? kb : (kb < R - 1 ? kb + 1 : R - 1); // register block that holds index k+1
            if (tx == kmn) {
#pragma unroll
                for (int a = 0; a < R; a++) colb[buf ^ 1][ty + 16 * a] = a_[a][kq] - c.l[a] * c.y[kq];
            }
            if (ty == kmn) {
#pragma unroll
                for (int b = 0; b < R; b++) roww[buf ^ 1][tx + 16 * b] = w_[kq][b] - c.l[kq] * c.wk[b];
            }
        }
        BA_STAMP_SEG(0);
        __syncthreads();
        BA_STAMP_SEG(1);
        // 2. LDS reads of pivot k+1 (harmless stale data when k+1 == nb)
        const T dn = colb[buf ^ 1][kn & (NB - 1)];
#pragma unroll
        for (int a = 0; a < R; a++) n.l[a] = colb[buf ^ 1][ty + 16 * a];
#pragma unroll
        for (int b = 0; b < R; b++) { n.y[b] = colb[buf ^ 1][tx + 16 * b]; n.wk[b] = roww[buf ^ 1][tx + 16 * b]; }
        BA_STAMP_SEG(2);
        // 3. rank-1 updates of pivot k, interleaved with the reciprocal chain of pivot k+1
// Only the register blocks that can hold live entries are touched (kb is a compile-time constant of the
        // unrolled outer loop): A needs rows i > k and columns k < j <= i  ->  blocks a >= b >= kb;
        // W = L^-1 needs rows i > k and columns j <= k                     ->  blocks a >= kb, b <= kb.
        // (The loop is issue-bound: one wave issues an f64 FMA only every ~9 cycles, so skipped FMAs are time saved.)
#define BA_GRP_A(a) { _Pragma("unroll") for (int b = 0; b < R; b++) if (a >= kb && b >= kb && a >= b) a_[a < R ? a : 0][b] -= c.l[a < R ? a : 0] * c.y[b]; }
#define BA_GRP_W(a) { _Pragma("unroll") for (int b = 0; b < R; b++) if (a >= kb && b <= kb) w_[a < R ? a : 0][b] -= c.l[a < R ? a : 0] * c.wk[b]; }
        __builtin_amdgcn_sched_barrier(0);
        BA_GRP_A(0) BA_GRP_W(0)
        T r = ba_rcp_est(dn);
        if (R > 1) { BA_GRP_A(1) BA_GRP_W(1) }
        T e = ba_fnma1(dn, r);
        if (R > 2) { BA_GRP_A(2) }
        r = ba_fma_(e, r, r);
        if (R > 2) { BA_GRP_W(2) }
        if (sizeof(T) == 8) e = ba_fnma1(dn, r);
        if (R > 3) { BA_GRP_A(3) }
        if (sizeof(T) == 8) r = ba_fma_(e, r, r);
        if (R > 3) { BA_GRP_W(3) }
        __builtin_amdgcn_sched_barrier(0);
#undef BA_GRP_A
#undef BA_GRP_W
        BA_STAMP_SEG(3);
        n.inv = r;
#pragma unroll
        for (int a = 0; a < R; a++) n.l[a] = (ty + 16 * a > kn) ? n.l[a] * r : (T)0;
#pragma unroll
        for (int b = 0; b < R; b++) n.y[b] = (tx + 16 * b > kn) ? n.y[b] : (T)0;
        if (tx == km) {
#pragma unroll
            for (int a = 0; a < R; a++)
                if (ty + 16 * a > k) a_[a][kb] = c.l[a];
        }
        if (tx == km && ty == km) dinv[k] = c.inv;
        BA_STAMP_SEG(4);
    };
    if (tx == 0) {
#pragma unroll
        for (int a = 0; a < R; a++) colb[0][ty + 16 * a] = a_[a][0];
    }
    if (ty == 0) {
#pragma unroll
        for (int b = 0; b < R; b++) roww[0][tx + 16 * b] = w_[0][b];
    }
    __syncthreads();
    fetch0(pv0);
#pragma unroll
    for (int kb = 0; kb < R; kb++) {
#pragma unroll 1
        for (int km = 0; km < 16; km += 2) { // a real loop: 64 specialised pivot bodies would thrash the instruction cache
            if (16 * kb + km >= nb) break;
            pivot(kb, km, pv0, pv1);
            if (16 * kb + km + 1 >= nb) break;
            pivot(kb, km + 1, pv1, pv0);
        }
    }
    __syncthreads();
#ifdef BA_STAMP
    if (blockIdx.x == 0 && (tid & 63) == 0)
        for (int q = 0; q < 6; q++) ba_stamp_acc[8 * (tid >> 6) + q] = st_acc[q];
#endif
    // publish W (LDS for the GEMM below, global for the backward sweep) and the factored block
#pragma unroll
    for (int a = 0; a < R; a++)
#pragma unroll
        for (int b = 0; b < R; b++) {
            const int i = ty + 16 * a, j = tx + 16 * b;
            const T wv = (j <= i && i < nb) ? w_[a][b] : (T)0;
            Wl[i][j] = wv;
            if (blockIdx.x == 0) {
                Winv[i * NB + j] = wv;
                if (j <= i && j < nb) S[(size_t)(p0 + j) * ld + p0 + i] = a_[a][b];
            }
        }
    __syncthreads();
    // rows below: Y^T = W X^T on the matrix cores; wave w owns 16 rows
    const int lane = tid & 63, wv_ = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int r0 = p0 + NB + 64 * blockIdx.x + 16 * wv_;
    if (r0 >= nrows || nb < NB) return;
    typename ba_acc<T>::type acc[R];
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[t][v] = 0;
#pragma unroll
    for (int kk = 0; kk < NB / 4; kk++) {
        const T xb = S[(size_t)(p0 + 4 * kk + lk) * ld + r0 + li]; // B[k][n] = X[n][k]
#pragma unroll
        for (int t = 0; t < R; t++) {
            if (kk <= 4 * t + 3) { // W is lower triangular: W[j][k] = 0 for k > j
                const T wa = Wl[16 * t + li][4 * kk + lk]; // A[j][k]
                acc[t] = ba_mfma(wa, xb, acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int j = 16 * t + ba_crow<T>(lk, v); // column of the panel
            const T yv = acc[t][v];
            Wp[(size_t)j * ld + r0 + li] = yv;
            S[(size_t)(p0 + j) * ld + r0 + li] = yv * dinv[j];
        }
}

// Trailing update.  64 x 64 tile per workgroup; wave w owns a 32 x 32 quadrant (2 x 2 accumulators: two A and two B
// fragments feed four MFMAs).  The MFMA computes the TRANSPOSED tile  C^T[j][i] -= sum_k L[j][k] Y[i][k]  so that the
// 16-wide "column" index of the C/D fragment runs along the rows of S (contiguous in the column-major matrix): every
// accumulator load / store instruction touches 4 columns x 128 contiguous bytes.  Operands come straight from L2 (the
// 64-wide panel is a few MB at most); all 8 k-steps of a half are in flight at once to cover the L2 latency.
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_update(int nrows, int ncols, int ld, int p0, T *__restrict__ S, const T *__restrict__ Wp)
{
    const int p1 = p0 + NB;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = p1 + 64 * ti + 32 * (w >> 1), col0 = p1 + 64 * tj + 32 * (w & 1);
    if (row0 >= nrows || col0 >= ncols) return;
    if (ti == tj && col0 > row0) return; // strictly upper quadrant of a diagonal tile
    const int li = lane & 15, lk = lane >> 4;
    // acc[t][u]: rows of C^T = columns col0 + 16 t + crow of S, cols of C^T = rows row0 + 16 u + li of S
    typename ba_acc<T>::type acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 4; v++)
                acc[t][u][v] = S[(size_t)(col0 + 16 * t + ba_crow<T>(lk, v)) * ld + row0 + 16 * u + li];
    constexpr int CH = (NB / 4 < 8) ? NB / 4 : 8; // k-steps whose operands are in flight together
#pragma unroll
    for (int half = 0; half < (NB / 4) / CH; half++) {
        T a[CH][2], b[CH][2];
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int kk = CH * half + q;
#pragma unroll
            for (int t = 0; t < 2; t++) a[q][t] = -S[(size_t)(p0 + 4 * kk + lk) * ld + col0 + 16 * t + li]; // A[j][k] = L[j][k]
#pragma unroll
            for (int u = 0; u < 2; u++) b[q][u] = Wp[(size_t)(4 * kk + lk) * ld + row0 + 16 * u + li];        // B[k][i] = Y[i][k]
        }
#pragma unroll
        for (int q = 0; q < CH; q++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < 2; u++) acc[t][u] = ba_mfma(a[q][t], b[q][u], acc[t][u]);
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 4; v++)
                S[(size_t)(col0 + 16 * t + ba_crow<T>(lk, v)) * ld + row0 + 16 * u + li] = acc[t][u][v];
}

// Backward sweep L^T x = z, right-looking, one launch per block column (p0 descending).  z lives in row zrow of S.
// Every workgroup first finishes the unknowns of block p0 (x_p = W_p^T z_p, a 64x64 GEMV, redundantly), workgroup 0
// publishes them, then each wave eliminates them from its share of the earlier unknowns:
//   z_c -= sum_r L(p0 + r, c) x(p0 + r)    (64-lane coalesced read down column c + shuffle sum).
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_backstep(int ncols, int ld, int zrow, int p0, T *__restrict__ S,
                                                       const T *__restrict__ Winv, T *__restrict__ x)
{
    __shared__ T zs[NB], xs[NB], part[256 / NB][NB];
    const int tid = threadIdx.x;
    const int nb = min(NB, ncols - p0);
    if (tid < NB) zs[tid] = (tid < nb) ? S[(size_t)(p0 + tid) * ld + zrow] : (T)0;
    __syncthreads();
    {
        constexpr int NQ = 256 / NB, KQ = NB / NQ; // NQ partial sums of KQ terms per unknown
        const int j = tid % NB, q = tid / NB;
        T a = 0;
#pragma unroll
        for (int k = KQ * q; k < KQ * q + KQ; k++) a += Winv[k * NB + j] * zs[k];
        part[q][j] = a;
    }
    __syncthreads();
    if (tid < NB) {
        T xv = 0;
#pragma unroll
        for (int q = 0; q < 256 / NB; q++) xv += part[q][tid];
        xs[tid] = xv;
        if (blockIdx.x == 0 && tid < nb) x[p0 + tid] = xv;
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6;
    const T xl = (lane < NB) ? xs[lane < NB ? lane : 0] : (T)0;
    for (int c = blockIdx.x * 4 + w; c < p0; c += gridDim.x * 4) {
        T a = (lane < NB) ? S[(size_t)c * ld + p0 + lane] * xl : (T)0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) S[(size_t)c * ld + zrow] -= a;
    }
}

#endif
