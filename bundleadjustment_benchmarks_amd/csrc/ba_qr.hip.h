// ba_qr.hip.h -- the QRKIT symbol's right block: dense thin Householder QR of J2bot (src/Optimization/BAFunctor.h:99-102:
// BlockAngularSparseQR< J, BlockDiagonalSparseQR<ColPivHouseholderQR>, DenseBlockedThinQR<MatrixXX, NaturalOrdering, 4, true> >;
// README.md:14 "block diagonal QR on the left block, dense QR on the lower-right block").  QRKit itself is not vendored; what is
// restated is what that type says: after the per-point QR of the left block (k_elim_qr) the rows of Q^T [J_c ; 0] below each
// point's top three -- J2bot, (2K + 3M + D) x D with the camera sqrt(lambda) rows, dense storage -- are factored by Householder
// reflections and the camera step solves  min || J2bot y + qtb2 ||  through R y = -Q^T qtb2.  No normal equations: this is the
// one symbol whose reduced system is never squared (cond(J2bot) = sqrt(cond(S))), which is what it is there for in fp32.
//
// J2bot is built as (I - Q1 Q1^T) [A ; 0] per point -- all 2 k_j + 3 rows, i.e. the rows orthogonal to the point's thin Q1 up to
// an orthogonal row transform, which leaves R and the least-squares solution unchanged (the CPU oracle does the same,
// oracle/ba_oracle_impl.h: solve_reduced_qr).
//
// The QR is blocked by 32-column panels; a panel is factored as a TSQR tree so that no reflector ever needs a grid-wide
// reduction: level 1 cuts the rows into chunks of CH (256 fp32 / 128 fp64) that ONE WAVEFRONT factors in registers (Householder,
// column by column, reflectors stored in place below the diagonal of the chunk, the chunk's 32 x 32 R on its top rows); level
// L + 1 stacks the R's of NSB = CH / 32 level-L chunks (their top rows, in place: row stride CH * NSB^(L-1)) and factors the
// stack the same way -- its reflectors only have entries where the stacked triangles had them, so they fit in the triangles
// they annihilate and the lower-level reflectors underneath stay intact.  The trailing columns (and the right-hand side, which
// rides along as column D) receive the chunk reflectors level by level: one wavefront per (chunk, 32 columns) holds the tile in
// registers and applies the 32 reflectors one after the other.  No LDS tile and no barrier anywhere: round 2's first version
// (one workgroup per 1024-row chunk, tile in LDS, five barriers per reflector) spent 8 - 11 us per reflector waiting on LDS
// round trips and ran 26 ms per trial at config 3.
#ifndef BA_QR_HIP_H
#define BA_QR_HIP_H

#include <hip/hip_runtime.h>
#include "ba_mfma.hip.h"

#define BA_QR_PB 32 /* panel width = rows of a sub-block */
#define BA_QR_CW 8  /* trailing columns per wavefront of k_qr_apply */

template <typename T> struct ba_qr_cfg {
    static constexpr int NSB = sizeof(T) == 4 ? 8 : 4; // sub-blocks (of 32 rows) per chunk
    static constexpr int CH = BA_QR_PB * NSB;          // rows per chunk = 64 lanes x 4 (fp32) / 2 (fp64) rows: a CH x 32 tile is 128 registers per lane
};

// global row of local row l of chunk g: sub-block s = l / 32 starts at row0 + (g NSB + s) stride, stride = 32 at level 1
template <typename T> __device__ __forceinline__ size_t ba_qr_row(int row0, int g, int l, long long stride)
{
    return (size_t)row0 + (size_t)((long long)g * ba_qr_cfg<T>::NSB + (l >> 5)) * (size_t)stride + (size_t)(l & 31);
}

// ---- J2bot ------------------------------------------------------------------------------------------------------------
// One thread per observation ia (point j, camera a): the 9 columns of camera a in all rows of point j:
//   observation rows of ib:  delta(ia, ib) A_ib - Q1_ib Z_ia^T   (2 x 9)      Z_ia = R12_ia^T = A_ia^T Q1_ia (rec)
//   lambda rows of j:        - Q1lam_j Z_ia^T                    (3 x 9)
// and, by the first observation of the point, the right-hand side column D:  -(r_ib - Q1_ib q1) and +Q1lam q1  (q1 = -tvec).
// Row layout: point j with observations [b, e) owns rows 2 b + 3 j ... ; the camera sqrt(lambda) rows follow at 2 K + 3 M.
template <typename T>
__global__ __launch_bounds__(256) void k_qrkit_build(int K, int Ml, int D, const int *__restrict__ obs_cam, const int *__restrict__ obs_pt,
                                                     const int *__restrict__ pt_ptr, const T *__restrict__ Jc /* SoA [18][K] */,
                                                     const T *__restrict__ r /* SoA [2][K] */, const T *__restrict__ rec,
                                                     const T *__restrict__ q1obs /* [K][6] */, const T *__restrict__ q1lam /* [Ml][9] */,
                                                     const T *__restrict__ tvec /* SoA [3][Ml] = -q1 */, const T *__restrict__ lam,
                                                     T *__restrict__ A, size_t lda)
{
    const int ia = blockIdx.x * 256 + threadIdx.x;
    if (ia < D) A[(size_t)ia * lda + 2 * (size_t)K + 3 * (size_t)Ml + ia] = sqrt(*lam); // camera rows: sqrt(lambda) I_D, zero rhs
    if (ia >= K) return;
    const int j = obs_pt[ia], a = obs_cam[ia], b = pt_ptr[j], e = pt_ptr[j + 1];
    const size_t r0 = 2 * (size_t)b + 3 * (size_t)j;
    T Z[27];
#pragma unroll
    for (int q = 0; q < 27; q++) Z[q] = rec[(size_t)ia * BA_REC + q];
    T *colbase = A + (size_t)(9 * a) * lda;
    for (int ib = b; ib < e; ib++) {
        const T *Q = q1obs + 6 * (size_t)ib; // 2 x 3 row-major
        T q[6];
#pragma unroll
        for (int m = 0; m < 6; m++) q[m] = Q[m];
#pragma unroll
        for (int c = 0; c < 9; c++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                T v = (ib == ia) ? Jc[(size_t)(9 * rr + c) * K + ia] : (T)0;
                v -= q[3 * rr] * Z[3 * c] + q[3 * rr + 1] * Z[3 * c + 1] + q[3 * rr + 2] * Z[3 * c + 2];
                colbase[(size_t)c * lda + r0 + 2 * (size_t)(ib - b) + rr] = v;
            }
    }
    const T *Ql = q1lam + 9 * (size_t)j; // 3 x 3 row-major: lambda row rr, column m
    const size_t rl = r0 + 2 * (size_t)(e - b);
#pragma unroll
    for (int c = 0; c < 9; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
            colbase[(size_t)c * lda + rl + rr] = -(Ql[3 * rr] * Z[3 * c] + Ql[3 * rr + 1] * Z[3 * c + 1] + Ql[3 * rr + 2] * Z[3 * c + 2]);
    if (ia == b) { // right-hand side of the point: -(qtb2) = -( [r ; 0] - Q1 q1 ),  q1 = -t
        const T t0 = -tvec[j], t1 = -tvec[(size_t)Ml + j], t2 = -tvec[2 * (size_t)Ml + j];
        T *rhs = A + (size_t)D * lda;
        for (int ib = b; ib < e; ib++) {
            const T *Q = q1obs + 6 * (size_t)ib;
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
                rhs[r0 + 2 * (size_t)(ib - b) + rr] = -(r[(size_t)rr * K + ib] - (Q[3 * rr] * t0 + Q[3 * rr + 1] * t1 + Q[3 * rr + 2] * t2));
        }
#pragma unroll
        for (int rr = 0; rr < 3; rr++) rhs[rl + rr] = Ql[3 * rr] * t0 + Ql[3 * rr + 1] * t1 + Ql[3 * rr + 2] * t2;
    }
}

// ---- one chunk of a TSQR level: Householder QR of its rows of the panel, in registers --------------------------------------
// Lane l holds rows l, l + 64, ... of the chunk (RPL = CH / 64 of them); the 32 panel columns are dealt to the four wavefronts of
// the workgroup cyclically.  Step j: the owner of column j forms the reflector (norm below the pivot by a wave reduction, the
// scalars redundantly in every lane), hands it to the others through LDS and retires the column to memory (R entries above the
// pivot, beta on it, v below), shifting its registers left by one column so that its next column is at position 0 again -- the
// loop body is the same for every j; then every wavefront updates its own columns, one wave reduction per column for v . a_c.
// level 1: the chunk's rows are dense; level > 1: every sub-block of 32 rows is an upper triangle (the R of a lower-level chunk) --
// entries below a sub-block's diagonal are read as zero and never written (the lower level's reflectors live there).
template <typename T>
__global__ __launch_bounds__(256) void k_qr_chunk(T *__restrict__ A, size_t lda, int c0, int bw, int row0, int level, long long stride, int nsb_total,
                                                  T *__restrict__ tau /* [chunks][32] */, int nch)
{
    // One workgroup per chunk, wave w owns the panel columns c with (c & 3) == w (eight of them, CW): the column that step j
    // retires is always at register position 0 of its owner.  The owner forms the reflector (norm, scalars), leaves it in LDS
    // (double-buffered: one barrier per step) and retires its column; every wave then updates its own columns.
    constexpr int NSB = ba_qr_cfg<T>::NSB, CH = ba_qr_cfg<T>::CH, RPL = CH / 64, CW = BA_QR_PB / 4;
    __shared__ T vs[2][CH];
    __shared__ T tj_s[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = blockIdx.x;
    const int nsb = min(NSB, nsb_total - g * NSB), rows = BA_QR_PB * nsb;
    T a[RPL][CW];
    size_t grow[RPL];
#pragma unroll
    for (int e = 0; e < RPL; e++) {
        const int l = lane + 64 * e;
        grow[e] = ba_qr_row<T>(row0, g, l, stride);
#pragma unroll
        for (int q = 0; q < CW; q++) {
            const int c = 4 * q + wv;
            a[e][q] = (c < bw && l < rows && (level == 1 || (l & 31) <= c)) ? A[(size_t)(c0 + c) * lda + grow[e]] : (T)0;
        }
    }
    for (int j = 0; j < bw; j++) {
        if ((j & 3) == wv) { // (wave-uniform)
            T part = 0;
#pragma unroll
            for (int e = 0; e < RPL; e++) part += (lane + 64 * e > j) ? a[e][0] * a[e][0] : (T)0;
            const T x2 = ba_wave_sum_all<T>(part);
            const T alpha = __shfl(a[0][0], j, 64); // row j lives in lane j, e = 0 (j < 32)
            T tj = 0, sc = 0, beta = alpha;
            if (x2 != (T)0) { // (a column that is already zero below its pivot keeps the identity reflector)
                beta = sqrt(alpha * alpha + x2);
                if (alpha > (T)0) beta = -beta;
                tj = (beta - alpha) / beta;
                sc = (T)1.0 / (alpha - beta);
            }
            if (lane == 0) { tau[(size_t)g * BA_QR_PB + j] = tj; tj_s[j & 1] = tj; }
#pragma unroll
            for (int e = 0; e < RPL; e++) {
                const int l = lane + 64 * e;
                const T ve = l > j ? a[e][0] * sc : (l == j ? (T)1 : (T)0);
                vs[j & 1][l] = ve;
                const T keep = l > j ? ve : (l == j ? beta : a[e][0]);
                if (l < rows && (level == 1 || (l & 31) <= j)) A[(size_t)(c0 + j) * lda + grow[e]] = keep; // retire column j
#pragma unroll
                for (int q = 0; q + 1 < CW; q++) a[e][q] = a[e][q + 1];
                a[e][CW - 1] = 0;
            }
        }
        __syncthreads();
        const T tj = tj_s[j & 1];
        T v[RPL];
#pragma unroll
        for (int e = 0; e < RPL; e++) v[e] = vs[j & 1][lane + 64 * e];
        // w_c = tau (v . a_c) for this wave's columns behind j, a_c -= v w_c  (v is 1 on row j, zero above)
#pragma unroll
        for (int q = 0; q < CW; q++) {
            T pd = 0;
#pragma unroll
            for (int e = 0; e < RPL; e++) pd += v[e] * a[e][q];
            const T w = tj * ba_wave_sum_all<T>(pd);
#pragma unroll
            for (int e = 0; e < RPL; e++) a[e][q] -= v[e] * w;
        }
    }
}

// ---- the reflectors of one chunk applied to CW trailing columns: one wavefront per (chunk, column strip), in registers ----------
// Reflector j of the chunk: 1 at local row j, zero above; below: level 1 -- the stored panel column; level > 1 -- in every
// sub-block behind the first only the rows t <= j (the triangle it annihilated).  The next reflector's entries are requested
// while the current one is applied.  Eight columns per wavefront: 32 registers of tile, eight wavefronts per SIMD hide the L2
// latency of the reflector loads, and a task is 32 x 8 short dependent chains (6 us) -- the upper levels of the tree, where a
// launch holds a handful of tasks, take as long as one task.
template <typename T>
__global__ __launch_bounds__(256) void k_qr_apply(T *__restrict__ A, size_t lda, int c0, int bw, int row0, int level, long long stride, int nsb_total,
                                                  const T *__restrict__ tau, int col0, int col1, int nch, int nct)
{
    constexpr int NSB = ba_qr_cfg<T>::NSB, CH = ba_qr_cfg<T>::CH, RPL = CH / 64, CW = BA_QR_CW;
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nch * nct) return;
    const int g = wid % nch, ct = wid / nch; // (neighbouring wavefronts share a column strip and walk neighbouring chunks)
    const int cb = col0 + CW * ct, ncol = min(CW, col1 - cb);
    const int nsb = min(NSB, nsb_total - g * NSB), rows = BA_QR_PB * nsb;
    T b[RPL][CW];
    size_t grow[RPL];
#pragma unroll
    for (int e = 0; e < RPL; e++) {
        const int l = lane + 64 * e;
        grow[e] = ba_qr_row<T>(row0, g, l, stride);
#pragma unroll
        for (int c = 0; c < CW; c++) b[e][c] = (c < ncol && l < rows) ? A[(size_t)(cb + c) * lda + grow[e]] : (T)0;
    }
    auto vload = [&](int j, T (&v)[RPL]) {
#pragma unroll
        for (int e = 0; e < RPL; e++) {
            const int l = lane + 64 * e;
            T x = 0;
            if (l == j) x = 1;
            else if (l > j && l < rows && (level == 1 ? true : ((l >> 5) > 0 && (l & 31) <= j))) x = A[(size_t)(c0 + j) * lda + grow[e]];
            v[e] = x;
        }
    };
    T vn[RPL];
    vload(0, vn);
    for (int j = 0; j < bw; j++) {
        const T tj = tau[(size_t)g * BA_QR_PB + j];
        T v[RPL];
#pragma unroll
        for (int e = 0; e < RPL; e++) v[e] = vn[e];
        if (j + 1 < bw) vload(j + 1, vn);
#pragma unroll
        for (int c = 0; c < CW; c++) {
            T pd = 0;
#pragma unroll
            for (int e = 0; e < RPL; e++) pd += v[e] * b[e][c];
            const T w = tj * ba_wave_sum_all<T>(pd);
#pragma unroll
            for (int e = 0; e < RPL; e++) b[e][c] -= v[e] * w;
        }
    }
#pragma unroll
    for (int e = 0; e < RPL; e++) {
        const int l = lane + 64 * e;
#pragma unroll
        for (int c = 0; c < CW; c++)
            if (c < ncol && l < rows) A[(size_t)(cb + c) * lda + grow[e]] = b[e][c];
    }
}

// ---- R y = (Q^T rhs)[0 : D): back substitution by one workgroup ------------------------------------------------------------
// R is the upper triangle of the first D rows of A, the transformed right-hand side column D.  Column-oriented: y_j =
// b_j / R_jj, then b_i -= R_ij y_j for i < j (the column of R is contiguous in memory).
template <typename T>
__global__ __launch_bounds__(256) void k_qr_backsolve(const T *__restrict__ A, size_t lda, int D, T *__restrict__ y)
{
    extern __shared__ unsigned char smem_raw[];
    T *b = reinterpret_cast<T *>(smem_raw);
    const int tid = threadIdx.x;
    for (int i = tid; i < D; i += 256) b[i] = A[(size_t)D * lda + i];
    __syncthreads();
    for (int j = D - 1; j >= 0; j--) {
        const T *col = A + (size_t)j * lda;
        const T yj = b[j] / col[j];
        __syncthreads();
        if (tid == 0) { b[j] = yj; y[j] = yj; }
        for (int i = tid; i < j; i += 256) b[i] -= col[i] * yj;
        __syncthreads();
    }
}

// Host side: QR of the (mrows x D) matrix A (+ rhs in column D) on `st`, then y = argmin || A y - rhs ||.
// A: lda >= mrows + 64 rows allocated and zero beyond mrows.  tau: room for (ceil(mrows / CH) + 2) * 32 scalars per level, 8 levels
// (CH = 256 / 128 rows: 181 633 rows are 5 levels in fp32, 7 in fp64).
// st2 != nullptr (with two events): the trailing updates run on st2 beside the panel's chunk chain -- level L + 1 of the chain only
// needs the panel's own R's from level L, not the trailing update of level L -- and the next panel waits for the last of them
// (fork / join by events: also valid inside a stream capture).  6.2 -> 5.3 ms per trial at config 3.  With look-ahead on top (every
// level's reflectors to the next panel's 32 columns first, on `st`, so that the next chain starts before the rest is done) it was
// 5.7 ms: five more launches of one task's latency each on the critical stream cost more than the overlap gives.
template <typename T>
inline void ba_qr_solve(hipStream_t st, T *A, size_t lda, int mrows, int D, T *tau, size_t tau_level_stride, T *y, hipStream_t st2 = nullptr,
                        hipEvent_t ev_chunk = nullptr, hipEvent_t ev_apply = nullptr)
{
    constexpr int NSB = ba_qr_cfg<T>::NSB;
    const bool two = st2 != nullptr && ev_chunk != nullptr && ev_apply != nullptr;
    for (int c0 = 0; c0 < D; c0 += BA_QR_PB) {
        const int bw = D - c0 < BA_QR_PB ? D - c0 : BA_QR_PB;
        const int col0 = c0 + bw, col1 = D + 1; // trailing columns incl. the right-hand side
        int nsb = (mrows - c0 + BA_QR_PB - 1) / BA_QR_PB; // 32-row blocks from the panel's first row down
        long long stride = BA_QR_PB;
        const int nct = (col1 - col0 + BA_QR_CW - 1) / BA_QR_CW;
        for (int level = 1;; level++) {
            const int nch = (nsb + NSB - 1) / NSB;
            T *tl = tau + (size_t)(level - 1) * tau_level_stride;
            hipLaunchKernelGGL((k_qr_chunk<T>), dim3(nch), dim3(256), 0, st, A, lda, c0, bw, c0, level, stride, nsb, tl, nch);
            if (nct > 0) {
                hipStream_t sa = st;
                if (two) {
                    (void)hipEventRecord(ev_chunk, st);
                    (void)hipStreamWaitEvent(st2, ev_chunk, 0);
                    sa = st2;
                }
                hipLaunchKernelGGL((k_qr_apply<T>), dim3((unsigned)(((long long)nch * nct + 3) / 4)), dim3(256), 0, sa, A, lda, c0, bw, c0, level, stride, nsb,
                                   (const T *)tl, col0, col1, nch, nct);
            }
            if (nch == 1) break;
            nsb = nch;
            stride *= NSB;
        }
        if (two && nct > 0) { // the next panel (and the back substitution) read what the trailing updates wrote
            (void)hipEventRecord(ev_apply, st2);
            (void)hipStreamWaitEvent(st, ev_apply, 0);
        }
    }
    hipLaunchKernelGGL((k_qr_backsolve<T>), dim3(1), dim3(256), sizeof(T) * (size_t)D, st, (const T *)A, lda, D, y);
}

#endif
