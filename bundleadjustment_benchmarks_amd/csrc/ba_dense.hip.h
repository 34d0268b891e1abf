// ba_dense.hip.h -- K6: dense LDL^T of the reduced camera matrix and the triangular solves, gfx950.
//
// Stands in for Eigen::SimplicialLDLT on J2bot^T J2bot (src/Optimization/BAFunctor.h:106,
// src/Eigen_ext/BacktrackLevMarqQRChol.h:339-341) and, for the CHOLESKY symbol, for the camera part of the
// LDL^T of the whole J^T J + lambda I (src/Eigen_ext/BacktrackLevMarqCholesky.h:156,274-282).  The reduced matrix
// is 85-100 % block-dense, so it is factored as a dense matrix: right-looking, NB-wide block columns,
//   k_ldlt_panel  : every workgroup factors the NB x NB diagonal block in LDS (redundantly -- it is the critical
//                   path and a broadcast would cost a kernel boundary), then forward-substitutes its own rows
//   k_ldlt_update : trailing update S_ij -= (L D)_i L_j^T on the matrix cores (v_mfma_f64_16x16x4_f64 /
//                   v_mfma_f32_16x16x4_f32) -- the one true contraction of the LM trial
// No pivoting, no square roots: D keeps the sign of a pivot, like SimplicialLDLT.
// The right-hand side rides along as the extra matrix row D: after the factorisation that row holds
// D^-1 L^-1 b, so only the backward sweep L^T x = z remains (k_ldlt_backstep, one launch per block column).
#ifndef BA_DENSE_HIP_H
#define BA_DENSE_HIP_H

#include <hip/hip_runtime.h>

typedef double ba_d4 __attribute__((ext_vector_type(4)));
typedef float ba_f4 __attribute__((ext_vector_type(4)));

// Factor the diagonal block held in LDS A[NB][NB+1] (lower triangle valid) for pivots k < nb.
// On exit: A[i][k] = L(i,k) for i > k, A[k][k] = D(k).  All threads of the 256-thread block take part.
template <typename T, int NB> __device__ __forceinline__ void ldlt_diag_block(T (*A)[NB + 1], T *ycol, int nb)
{
    const int tid = threadIdx.x;
    for (int k = 0; k < nb; k++) {
        __syncthreads();
        const T d = A[k][k];
        if (tid < NB && tid > k) {
            const T y = A[tid][k];
            ycol[tid] = y;
            A[tid][k] = y / d;
        }
        __syncthreads();
        const int w = NB - k - 1;
        for (int idx = tid; idx < w * w; idx += 256) {
            const int i = k + 1 + idx / w, j = k + 1 + idx % w;
            if (j <= i && j < nb) A[i][j] -= ycol[i] * A[j][k];
        }
    }
    __syncthreads();
}

// Panel step for block column p0: rows [p0, nrows), pivots [p0, min(p0 + NB, ncols)).
// Wp (ld x NB, column-major) receives Y = L D for the rows below the diagonal block (operand of the update).
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_panel(int nrows, int ncols, int ld, int p0, T *__restrict__ S, T *__restrict__ Wp)
{
    __shared__ T A[NB][NB + 1];
    __shared__ T ycol[NB];
    const int tid = threadIdx.x;
    const int nb = min(NB, ncols - p0);
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx % NB, j = idx / NB;
        A[i][j] = (j <= i) ? S[(size_t)(p0 + j) * ld + p0 + i] : (T)0;
    }
    ldlt_diag_block<T, NB>(A, ycol, nb);
    if (blockIdx.x == 0) {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int i = idx % NB, j = idx / NB;
            if (j <= i && j < nb) S[(size_t)(p0 + j) * ld + p0 + i] = A[i][j];
        }
    }
    const int r = p0 + NB + blockIdx.x * 256 + tid;
    if (r < nrows) {
        T y[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) y[c] = (c < nb) ? S[(size_t)(p0 + c) * ld + r] : (T)0;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            if (c < nb) {
                T a = y[c];
#pragma unroll
                for (int m = 0; m < c; m++) a -= y[m] * A[c][m];
                y[c] = a;
            }
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            Wp[(size_t)c * ld + r] = y[c];
            if (c < nb) S[(size_t)(p0 + c) * ld + r] = y[c] / A[c][c];
        }
    }
}

// Trailing update with the matrix cores.  64 x 64 tile per workgroup (4 waves, each 16 rows x 64 columns =
// four 16x16 accumulators sharing one A fragment).  Fragment maps (cdna_hip_programming.md s3):
//   A[i][k]: lane l holds i = l & 15, k = l >> 4;  B[k][j]: lane l holds k = l >> 4, j = l & 15
//   f64 C/D: col = l & 15, row = (l >> 4) + 4 v;   f32 C/D: col = l & 15, row = 4 (l >> 4) + v
// Operands come straight from L2 (the NB-wide panel is at most a few MB); no LDS staging is needed at this
// arithmetic intensity because every 8-byte operand element feeds a 64-cycle MFMA.
template <int NB>
__global__ __launch_bounds__(256) void k_ldlt_update_f64(int nrows, int ncols, int ld, int p0, double *__restrict__ S,
                                                         const double *__restrict__ Wp)
{
    const int p1 = p0 + NB;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int row0 = p1 + 64 * ti, col0 = p1 + 64 * tj;
    if (row0 >= nrows || col0 >= ncols) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int rbase = row0 + 16 * w;
    ba_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[t][v] = S[(size_t)(col0 + 16 * t + li) * ld + rbase + lk + 4 * v];
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; kk++) {
        const double a = -Wp[(size_t)(4 * kk + lk) * ld + rbase + li];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double b = S[(size_t)(p0 + 4 * kk + lk) * ld + col0 + 16 * t + li];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) S[(size_t)(col0 + 16 * t + li) * ld + rbase + lk + 4 * v] = acc[t][v];
}

template <int NB>
__global__ __launch_bounds__(256) void k_ldlt_update_f32(int nrows, int ncols, int ld, int p0, float *__restrict__ S,
                                                         const float *__restrict__ Wp)
{
    const int p1 = p0 + NB;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int row0 = p1 + 64 * ti, col0 = p1 + 64 * tj;
    if (row0 >= nrows || col0 >= ncols) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int rbase = row0 + 16 * w;
    ba_f4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[t][v] = S[(size_t)(col0 + 16 * t + li) * ld + rbase + 4 * lk + v];
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; kk++) {
        const float a = -Wp[(size_t)(4 * kk + lk) * ld + rbase + li];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float b = S[(size_t)(p0 + 4 * kk + lk) * ld + col0 + 16 * t + li];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) S[(size_t)(col0 + 16 * t + li) * ld + rbase + 4 * lk + v] = acc[t][v];
}

// Backward sweep L^T x = z, right-looking, one launch per block column (p0 descending).  z lives in row zrow of S
// (the augmented rhs row).  Every workgroup first finishes the NB unknowns of block p0 in LDS (redundantly, same
// reason as in the panel kernel), workgroup 0 publishes them to x, then each wave eliminates them from its share of
// the earlier unknowns:  z_c -= sum_r L(p0 + r, c) x(p0 + r)  -- a 64-lane coalesced read down column c + a shuffle sum.
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_backstep(int ncols, int ld, int zrow, int p0, T *__restrict__ S, T *__restrict__ x)
{
    __shared__ T xs[NB];
    const int tid = threadIdx.x;
    const int nb = min(NB, ncols - p0);
    if (tid < NB) xs[tid] = (tid < nb) ? S[(size_t)(p0 + tid) * ld + zrow] : (T)0;
    __syncthreads();
    for (int k = nb - 1; k > 0; k--) {
        const T xk = xs[k];
        if (tid < k) xs[tid] -= S[(size_t)(p0 + tid) * ld + p0 + k] * xk;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid < nb) x[p0 + tid] = xs[tid];
    const int lane = tid & 63, w = tid >> 6;
    for (int c = (blockIdx.x * 4 + w); c < p0; c += gridDim.x * 4) {
        T a = 0;
        for (int rr = lane; rr < nb; rr += 64) a += S[(size_t)c * ld + p0 + rr] * xs[rr];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) S[(size_t)c * ld + zrow] -= a;
    }
}

#endif
