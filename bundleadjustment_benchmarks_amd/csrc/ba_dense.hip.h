// ba_dense.hip.h -- K6: dense LDL^T of the reduced camera matrix and the triangular solves, gfx950.
//
// Stands in for Eigen::SimplicialLDLT on J2bot^T J2bot (src/Optimization/BAFunctor.h:106,
// src/Eigen_ext/BacktrackLevMarqQRChol.h:339-341) and, for the CHOLESKY symbol, for the camera part of the
// LDL^T of the whole J^T J + lambda I (src/Eigen_ext/BacktrackLevMarqCholesky.h:156,274-282).  The reduced matrix
// is 85-100 % block-dense, so it is factored as a dense matrix: right-looking, 64-wide block columns.
//
//   k_ldlt_panel    panel step of one block column (first block column, small matrices): every workgroup factors the
//                   64x64 diagonal block itself (it is the critical path; a broadcast would cost a kernel boundary) in
//                   four 16-wide sub-panels -- one wave runs the pivot loop of a 16x16 tile in registers, a second one
//                   builds the tile's inverse one pivot behind, the others do MFMA work beside them (ba_panel_body) --
//                   and keeps W = L11^-1; the rows below then need Y = A21 W^T, a GEMM on the matrix cores.
//   k_ldlt_step     fused step with look-ahead: panel of block column p, the previous panel's update of block column p
//                   (diagonal block inside the panel workgroups, their rows on workgroups of their own) and the rest of
//                   the previous panel's trailing update, in ONE launch per block column.
//   k_ldlt_update   stand-alone trailing update S_ij -= (L D)_i L_j^T on the matrix cores (v_mfma_f64_16x16x4_f64 /
//                   v_mfma_f32_16x16x4_f32) -- the one true contraction of the LM trial.
//   k_ldlt_backflow backward sweep L^T x = z in one launch (data flow between workgroups), using the stored W (x_p = W_p^T z_p);
//   k_ldlt_backpair / k_ldlt_backstep: the same with a launch per two / one block column(s).
//
// No pivoting, no square roots: D keeps the sign of a pivot, like SimplicialLDLT.  The right-hand side rides along
// as the extra matrix row `zrow` = D: after the factorisation that row holds D^-1 L^-1 b.
//
// MFMA fragment maps (cdna_hip_programming.md s3): A[i][k]: lane l holds i = l & 15, k = l >> 4; B[k][j]: lane l holds
// k = l >> 4, j = l & 15; C/D: col = l & 15, row = (l >> 4) + 4 v for f64 and 4 (l >> 4) + v for f32.
#ifndef BA_DENSE_HIP_H
#define BA_DENSE_HIP_H

#include <hip/hip_runtime.h>

#include "ba_mfma.hip.h"

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

#define BA_NB 64
// Bounds of the in-launch hand-off waits (a wait that runs out is an ERROR, reported through the device error word).
#define BA_FLAG_SPINS (1 << 22)
#define BA_SWEEP_SPINS (1 << 24)
// (BA_DEVERR_*: ba_mfma.hip.h)
// compile-time list of (up to four) tile indices
template <int N, int T0, int T1, int T2, int T3> struct ba_tiles {
    static constexpr int n = N;
    static __device__ constexpr int t(int u) { return u == 0 ? T0 : u == 1 ? T1 : u == 2 ? T2 : T3; }
};

// Diagnostic build only (-DBA_STAMP, scripts/bench_dense.hip): cycle stamps of the pivot loop's segments.
#ifdef BA_STAMP
__device__ long long ba_stamp_acc[8 * 8];
__device__ long long ba_stamp_own[4 * 4]; // own work of wave w in pivot-loop phase s: [4 * w + s]
__device__ long long ba_stamp_piv[4 * 4]; // phase s: factor wave has its tile [4 * s], has done its pivots [+ 1]
#define BA_STAMP_PIV(j) { unsigned long long t_; BA_STAMP_GET(t_); if (blk == 0 && lane == 0) ba_stamp_piv[4 * s + j] = (long long)(t_ - st_t0); }
#define BA_STAMP_DECL unsigned long long st_t0 = 0, st_t1 = 0; long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BA_STAMP_OWN(i) { BA_STAMP_GET(st_t1); st_acc[i] += (long long)(st_t1 - st_t0); if (blk == 0 && (threadIdx.x & 63) == 0) ba_stamp_own[4 * (threadIdx.x >> 6) + s] = (long long)(st_t1 - st_t0); } /* own work before a barrier */
#define BA_STAMP_PRO0 unsigned long long st_p0; BA_STAMP_GET(st_p0);
#define BA_STAMP_PRO unsigned long long st_p1; BA_STAMP_GET(st_p1); const long long st_pro = (long long)(st_p1 - st_p0);
#define BA_STAMP_FLUSH if (blk == 0 && (threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 7; q_++) ba_stamp_acc[8 * (threadIdx.x >> 6) + q_] = st_acc[q_]; ba_stamp_acc[8 * (threadIdx.x >> 6) + 7] = st_pro; }
#define BA_STAMP_GET(v) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define BA_STAMP_SEG(i) { BA_STAMP_GET(st_t1); st_acc[i] += (long long)(st_t1 - st_t0); st_t0 = st_t1; }
#else
#define BA_STAMP_DECL
#define BA_STAMP_OWN(i)
#define BA_STAMP_PRO
#define BA_STAMP_PRO0
#define BA_STAMP_FLUSH
#define BA_STAMP_GET(v)
#define BA_STAMP_SEG(i)
#define BA_STAMP_PIV(j)
#endif

// (ba_wave_lds_sync / ba_wave_lds_order: ba_mfma.hip.h)
// Lane K of every row of 16 lanes broadcast to that row (DPP row_newbcast): no LDS round trip.
template <int K> __device__ __forceinline__ double ba_rowbcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, true); } // ONE v_mov_b64_dpp
template <int K> __device__ __forceinline__ float ba_rowbcast(float v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, true); }
template <typename T> __device__ __forceinline__ T ba_rowbcast_k(T v, int k) // k is a constant after unrolling
{
    switch (k) {
    case 0: return ba_rowbcast<0>(v); case 1: return ba_rowbcast<1>(v); case 2: return ba_rowbcast<2>(v); case 3: return ba_rowbcast<3>(v);
    case 4: return ba_rowbcast<4>(v); case 5: return ba_rowbcast<5>(v); case 6: return ba_rowbcast<6>(v); case 7: return ba_rowbcast<7>(v);
    case 8: return ba_rowbcast<8>(v); case 9: return ba_rowbcast<9>(v); case 10: return ba_rowbcast<10>(v); case 11: return ba_rowbcast<11>(v);
    case 12: return ba_rowbcast<12>(v); case 13: return ba_rowbcast<13>(v); case 14: return ba_rowbcast<14>(v); default: return ba_rowbcast<15>(v);
    }
}

#ifndef BA_SENTINEL_DEFINED
#define BA_SENTINEL_DEFINED
// A bit pattern no arithmetic produces (hardware NaNs are canonical): marks "not written yet" in a hand-off buffer.
template <typename T> __device__ __forceinline__ T ba_sentinel();
template <> __device__ __forceinline__ double ba_sentinel<double>() { return __hiloint2double(-1, -1); }
template <> __device__ __forceinline__ float ba_sentinel<float>() { return __int_as_float(-1); }
__device__ __forceinline__ bool ba_is_sentinel(double v) { return __double2hiint(v) == -1 && __double2loint(v) == -1; }
__device__ __forceinline__ bool ba_is_sentinel(float v) { return __float_as_int(v) == -1; }
#endif

__device__ __forceinline__ double ba_readlane(double v, int lane)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
__device__ __forceinline__ float ba_readlane(float v, int lane)
{
    union { float f; int i; } u;
    u.f = v;
    u.i = __builtin_amdgcn_readlane(u.i, lane);
    return u.f;
}

template <typename T, int NB, bool TOLDS, bool STSC = false>
__device__ __forceinline__ void ba_update_tile(int ld, int p0, int row0t, int col0t, bool lower, T *__restrict__ S,
                                               const T *__restrict__ Wp, T (*Cl)[NB + 1]);
template <typename T, int NB, bool TOLDS, bool STSC = false>
__device__ __forceinline__ void ba_update_quad(int ld, int p0, int row0t, int col0t, bool lower, T *__restrict__ S,
                                               const T *__restrict__ Wp, T (*Cl)[NB + 1], int quad);
template <typename T, int NB>
__device__ __attribute__((noinline)) void ba_update_quad_call(int ld, int p0, int row0t, int col0t, T *S, const T *Wp, int quad);
template <typename T, int NB>
__device__ __forceinline__ void ba_update_macro(int nrows, int ncols, int ld, int pA, int row0, int col0, T *__restrict__ S,
                                                const T *__restrict__ W1, const T *__restrict__ W2, T *__restrict__ As, T *__restrict__ Bs);

// Panel step for block column p0: rows [p0, nrows), pivots [p0, min(p0 + 64, ncols)).
//   S    : in place; on exit the block column holds L (strictly lower) and D (diagonal)
//   Wp   : ld x 64, column-major: Y = L D for the rows below the diagonal block (operand of the trailing update)
//   Winv : 64 x 64 row-major: W = L11^-1 of this block (for the backward sweep)
// Grid: one workgroup per 64 rows below the diagonal block (at least one).
//
// Every workgroup factors the 64x64 diagonal block itself (it is the critical path; a broadcast would cost a kernel
// boundary).  The pivot recurrence d_k -> 1/d_k -> l_ik -> d_k+1 is pure latency on this machine (a dependent f64 FMA
// is ~32 cycles, one wave issues an f64 op every ~9 cycles), so the block is processed in four 16-wide sub-panels:
//   A1  wave 0 factors the 16x16 diagonal tile in registers (4 entries per lane) and carries its inverse along (the same
//       row operations on I, one pivot late, in the shadow of the next column exchange); the pivot is broadcast with
//       v_readlane so the reciprocal chain (estimate + Newton) starts before the LDS exchange of the column has finished;
//       the other three waves do the look-ahead update of the block, the deferred rank-16 updates and the W products;
//   A2  the tiles below (inside the 64x64 block) get Y = X W_ss^T on the matrix cores;
//   A3  the remaining tiles of the block get the rank-16 update on the matrix cores.
// W = L11^-1 (64x64) is then assembled from the four 16x16 inverses with MFMA products, and the rows below the diagonal
// block need Y = A21 W^T -- a GEMM, also on the matrix cores.
// flags != nullptr (k_ldlt_step<INL = true>): the look-ahead update of this workgroup's 64 rows below is done by ANOTHER
// workgroup of the same launch, which then stores `epoch` into flags[row block]; this workgroup waits for it just in front of
// the row GEMM (the update takes ~6 us, the diagonal block ~20: the wait is a formality, bounded in any case).
template <typename T, int NB, bool INL>
__device__ __forceinline__ void ba_panel_body(int nrows, int ncols, int ld, int p0, T *__restrict__ S, T *__restrict__ Wp,
                                              T *__restrict__ Winv, const T *__restrict__ Wprev, int blk, int nblk_panel,
                                              T (&Ad)[NB][NB + 1], T (&Wl)[NB][NB + 1], const int *flags = nullptr, int epoch = 0,
                                              T *errw = nullptr, int fault = 0 /* self-test: a short wait for an announcement that never comes */)
{
    static_assert(NB == 64, "the panel kernel is written for 64-wide block columns");
    // Ad: diagonal block, Ad[col][row]; lower tiles + full diagonal tiles are maintained.  Wl: W[row][col].  Both are declared by
    // the kernel: the macro-tile update of the same launch stages its operands in them (ba_update_macro).
    // Ys: unscaled sub-panel Y[col][row] (written in A2, read by the rank-16 updates, part of which run under the NEXT pivot
    // loop); Ts: per-wave scratch of the W tiles, kept in the strictly upper tiles of W's own image, which nobody reads.  ~79 KiB in total: two workgroups still fit a CU (the trailing-update
    // workgroups of the fused launch inherit the footprint).
    __shared__ T Ys[16][NB + 1];
#define BA_TS(sc, r, c) Wl[(r)][16 * ((sc) + 1) + (c)] /* scratch tile sc = the (never read) strictly upper tile (0, sc + 1) of W */
    __shared__ T colx4[4][16], dinv[NB], junkbuf[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int nb = min(NB, ncols - p0);
    BA_STAMP_PRO0
    // Tile fills: all global loads of a fill are issued before the first LDS store (a load -> store loop pays the L2 round
    // trip once per iteration: 16 x ~800 cycles for one 64 x 64 tile).
    // Only the ten lower 16 x 16 tiles are brought in (the strictly upper tiles of Ad are never read; the diagonal tiles are
    // mirrored into full ones): 20 KB per workgroup instead of 32 -- every panel workgroup pulls its own copy through one CU.
    constexpr int NF = 10;
    T fa[NF];
#pragma unroll
    for (int it = 0; it < NF; it++) {
        constexpr int TI[NF] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, TJ[NF] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
        const int r = 16 * TI[it] + (tid & 15), c = 16 * TJ[it] + (tid >> 4);
        fa[it] = S[(size_t)(p0 + min(r, c)) * ld + p0 + max(r, c)]; // mirror the lower triangle
    }
    // operands of the first look-ahead tiles (see below): requested together with the fill, used behind its barrier
    // (only in the one-workgroup-per-CU variant: the 64 registers would push the other one over 256 = one wave per SIMD)
    T t16a[NB / 4], t16b[NB / 4];
    if (INL && Wprev && wv < 1) {
        // (addresses by pointer increments: one 64-bit multiply per operand instead of one per load -- v_mad_i64_i32 is slow, and
        // these sit in front of the first loads of the kernel)
        const T *pa = S + (size_t)(p0 - NB + lk) * ld + p0 + li, *pb = Wprev + (size_t)lk * ld + p0 + 16 * wv + li;
        const size_t kstep = 4 * (size_t)ld;
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) {
            t16a[kk] = *pa; // (negated at use: a sign flip here makes the compiler wait for every older load before it issues the next)
            t16b[kk] = *pb;
            pa += kstep;
            pb += kstep;
        }
    }
    if (tid < NB) dinv[tid] = (T)0;
    typedef typename ba_acc<T>::type acc_t;
#pragma unroll
    for (int it = 0; it < NF; it++) {
        constexpr int TI[NF] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, TJ[NF] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
        const int r = 16 * TI[it] + (tid & 15), c = 16 * TJ[it] + (tid >> 4);
        Ad[c][r] = fa[it];
        // (W's image needs no clearing: every entry that is read -- the lower tiles and the full diagonal tiles -- is written first)
    }
    __syncthreads();
    // Look-ahead: the trailing update of the PREVIOUS block column (p0 - 64) runs in this same launch on other workgroups,
    // except for block column p0 itself, which this step needs now: every panel workgroup applies it to the diagonal block
    // (in LDS, redundantly) and to its own 64 rows below (in S).  The diagonal block is updated by 16 x 16 tiles in the
    // order the factorisation needs them: tile (0, 0) by wave 0 and (1, 0) by wave 1 right here (16 MFMAs each), the other
    // eight lower tiles by waves 2 and 3 while the first sub-panel is being factored; the 64 rows below during sub-panels
    // 1 and 2 (every later barrier waits for those stores, and the GEMM at the end reads them past L1).
    auto tile16_load = [&](int ti, int tj, T (&a)[NB / 4], T (&b)[NB / 4]) {
        const int pp = p0 - NB;
        const T *pa = S + (size_t)(pp + lk) * ld + p0 + 16 * tj + li, *pb = Wprev + (size_t)lk * ld + p0 + 16 * ti + li;
        const size_t kstep = 4 * (size_t)ld;
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) {
            a[kk] = *pa; // A[j][k] = L[j][k] (negated at use)
            b[kk] = *pb; // B[k][i] = Y[i][k]
            pa += kstep;
            pb += kstep;
        }
    };
    auto tile16_apply = [&](int ti, int tj, const T (&a)[NB / 4], const T (&b)[NB / 4]) {
        typename ba_acc<T>::type acc;
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = Ad[16 * tj + ba_crow<T>(lk, v)][16 * ti + li];
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) acc = ba_mfma(-a[kk], b[kk], acc);
#pragma unroll
        for (int v = 0; v < 4; v++) Ad[16 * tj + ba_crow<T>(lk, v)][16 * ti + li] = acc[v];
    };
    auto tile16 = [&](int ti, int tj) {
        T a[NB / 4], b[NB / 4];
        tile16_load(ti, tj, a, b);
        tile16_apply(ti, tj, a, b);
    };
    if (Wprev && wv < 1) { // (tile (1, 0) follows on wave 1 under the first pivot loop: nobody reads it before the barrier behind A1(0),
                           // and its 16 KB of operands no longer compete with the block fill for the CU's memory pipe)
        if (!INL) tile16_load(wv, 0, t16a, t16b);
        tile16_apply(wv, 0, t16a, t16b);
        ba_wave_lds_order(); // wave 0 reads its own tile in A1(0)
    }
    BA_STAMP_PRO
    // INL (the variant whose rows are updated by other workgroups): TWO panel workgroups per 64-row block, 32 rows each -- the
    // row GEMM behind the factorisation is MFMA-bound (40 x 64 cycles per wave for 64 rows) and the CUs are there.
    constexpr bool HALVES = INL;
    const int rblk = HALVES ? (blk >> 1) : blk;
    const int rown = p0 + NB + 64 * rblk;
    const bool own_rows = Wprev != nullptr && rown < nrows;
    BA_STAMP_DECL
    BA_STAMP_GET(st_t0);
    // Off-diagonal tile W_ts of W = L11^-1 (t > sc): W_ts = -W_tt sum_{u=sc}^{t-1} L_tu W_us, one wave, MFMA products.
    // Row t only needs rows < t of W, L_tu (final after A2 of sub-panel u) and W_tt (from A1 of sub-panel t), so row t is
    // built by the otherwise idle waves 1..3 while wave 0 runs A1 of sub-panel t + 1; only the last row is exposed.
    // w_sum: Ts[scratch] = sum_u L_tu W_us (needs neither W_tt nor the other tiles of row t); w_fin: W_ts = -W_tt Ts[scratch].
    auto w_sum = [&](int t, int sc, int scratch) {
        acc_t acc;
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = 0;
        for (int u = sc; u < t; u++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const T la = Ad[16 * u + 4 * kk + lk][16 * t + li];  // A[i][k] = L_tu[i][k]
                const T wb = Wl[16 * u + 4 * kk + lk][16 * sc + li]; // B[k][j] = W_us[k][j]
                acc = ba_mfma(la, wb, acc);
            }
#pragma unroll
        for (int v = 0; v < 4; v++) BA_TS(scratch, ba_crow<T>(lk, v), li) = acc[v];
        ba_wave_lds_order();
    };
    auto w_fin = [&](int t, int sc, int scratch) {
        acc_t acc2;
#pragma unroll
        for (int v = 0; v < 4; v++) acc2[v] = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const T wa = -Wl[16 * t + li][16 * t + 4 * kk + lk]; // A[i][k] = -W_tt[i][k]
            const T tb = BA_TS(scratch, 4 * kk + lk, li);        // B[k][j] = T[k][j]
            acc2 = ba_mfma(wa, tb, acc2);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) Wl[16 * t + ba_crow<T>(lk, v)][16 * sc + li] = acc2[v];
        ba_wave_lds_order();
    };
    auto w_tile = [&](int t, int sc, int scratch) { w_sum(t, sc, scratch); w_fin(t, sc, scratch); };
    // rank-16 update by the sub-panel at column cs (npv pivots) of tile (ti, tj) counted from the tile behind it:
    // C^T[j][i] -= sum_k L[j][k] Y[i][k]
    auto a3_tile = [&](int cs, int npv, int ti, int tj) {
        const int r0 = cs + 16 + 16 * ti, q0 = cs + 16 + 16 * tj;
        acc_t acc;
        T la[4], yb[4];
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = Ad[q0 + ba_crow<T>(lk, v)][r0 + li];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            la[kk] = Ad[cs + 4 * kk + lk][q0 + li]; // A[j][k] = L[j][k]
            yb[kk] = Ys[4 * kk + lk][r0 + li];      // B[k][i] = Y[i][k]
        }
        __builtin_amdgcn_sched_barrier(0); // twelve LDS reads in flight, then the MFMAs
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = ba_mfma((4 * kk + lk < npv) ? -la[kk] : (T)0, yb[kk], acc);
#pragma unroll
        for (int v = 0; v < 4; v++) Ad[q0 + ba_crow<T>(lk, v)][r0 + li] = acc[v];
    };
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
        const int c0 = 16 * s;
        const int np = min(16, nb - c0); // pivots in this sub-panel
        if (np <= 0) break;              // uniform
        // ---- A1: 16x16 diagonal tile; lane (i, q) owns row i, columns 4q .. 4q+3.
        // Wave 0 factors the tile (column k -> LDS, pivot by v_readlane, reciprocal, multipliers, rank-1 update) and applies
        // the same row operations to I one pivot late (W_ss = L_ss^-1: the multiplier is the lane's own register, row k of W
        // a DPP row broadcast, so the inverse costs no LDS traffic and hides in the wait for the next column exchange; a
        // second wave fed through LDS finished ~800 cycles behind the factor wave in every sub-panel).
        // D(k) stays in its register until the end (row k is never touched after pivot k); 1/D(k) is recomputed there by
        // the same instruction sequence, so the loop stores nothing but the multipliers.
        if (wv == 0) {
            const int i = li, q = lk;
            T a[4];
#pragma unroll
            for (int c = 0; c < 4; c++) a[c] = Ad[c0 + 4 * q + c][c0 + i];
            ba_wave_lds_order(); // the multipliers overwrite the tile: every lane has its entries first
            BA_STAMP_PIV(0)
            T *const junk = junkbuf + lane; // per-lane scratch slot
            T lprev = (T)0;
            T w[4];         // W_ss = L_ss^-1 in the same layout: the row operations of the factorisation applied to I, one pivot late
#pragma unroll
            for (int c = 0; c < 4; c++) w[c] = (4 * q + c == i) ? (T)1 : (T)0;
            // what is left of pivot k once its multipliers exist: L(., k) to its final place (zero on and above the diagonal),
            // and the same row operation on W -- the multiplier is the lane's own register and row k of W sits in lane k of every
            // row of 16 lanes (DPP broadcast), so this is eight VALU instructions that fill the wait for the NEXT exchange
            auto lstore = [&](int k) {
                *((q == (k >> 2)) ? &Ad[c0 + k][c0 + i] : junk) = lprev;
                T wk[4];
#pragma unroll
                for (int c = 0; c < 4; c++) wk[c] = ba_rowbcast_k(w[c], k);
#pragma unroll
                for (int c = 0; c < 4; c++) w[c] -= lprev * wk[c];
            };
            // One pivot = one LDS trip (column k out, five loads back, nothing waits for the store), the reciprocal (pivot by
            // v_readlane; hardware estimate + two Newton steps, kept in front of the wait for the loads: the wave issues in
            // order) and five FMAs.  The multipliers of pivot k - 1 are stored BEHIND the loads of pivot k, so that they
            // queue in the LDS while the wave computes and not in front of the exchange (148 against 192 cycles per pivot alone).
            auto pivot = [&](int k) {
                const int kq = k >> 2, kc = k & 3;
                colx4[q][i] = a[kc]; // column k is the kc-th register of the lanes with q == kq
                ba_wave_lds_order();
                const T lraw = colx4[kq][i];
                T y[4];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                ba_wave_lds_order();
                if (k > 0) lstore(k - 1);
                ba_wave_lds_order();
                const T dk = ba_readlane(a[kc], 16 * kq + k); // pivot: lane (i = k, q = kq)
                const T r = ba_rcp(dk);
                __builtin_amdgcn_sched_barrier(0);
                int iv = i; // (opaque copy: the comparison is made here, one v_cmp off the critical path; hoisted in front of the
                            // sub-panel loop the fifteen lane masks cost thirty SGPRs, their spills and a start-up delay for every wave)
                asm volatile("" : "+v"(iv));
                const T lm = (iv > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c]; // columns <= k are dead from here on
                lprev = l;
            };
            // pivot 15 has no rows below it in the tile: D(15) is final after pivot 14.  A full tile runs straight-line code;
            // only the last, partial tile of the last block column tests the pivot count.
            // (the last pivot's multipliers are flushed where its index is a compile-time constant: the DPP control is an immediate)
            if (np == 16) {
#pragma unroll
                for (int k = 0; k < 15; k++) pivot(k);
                BA_STAMP_PIV(1)
                lstore(14);
            } else {
#pragma unroll
                for (int k = 0; k < 15; k++)
                    if (k < np) { // uniform
                        pivot(k);
                        if (k + 1 == np || k == 14) lstore(k);
                    }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = 4 * q + c;
                Wl[c0 + i][c0 + j] = (j <= i) ? w[c] : (T)0;
            }
            { // D(i) sits in register i % 4 of lane (i, i / 4): selected, so that there is ONE reciprocal and not one per register
              // in four divergent branches (-200 cycles per sub-panel)
                const int ic = i & 3;
                const T dsel = (ic == 0) ? a[0] : (ic == 1) ? a[1] : (ic == 2) ? a[2] : a[3];
                if (q == (i >> 2) && i < np) {
                    Ad[c0 + i][c0 + i] = dsel; // D(i)
                    dinv[c0 + i] = ba_rcp(dsel);
                }
            }
        } else if (wv == 1) {
            // (The inverse rides on wave 0.)  Under the second pivot loop this wave takes the deferred rank-16 updates of the
            // tiles that get no look-ahead update in that phase -- (2,1), (3,2), (3,3) -- which leaves waves 2 and 3 one
            // rank-16 update + one look-ahead tile each.
            if (s == 0 && Wprev) tile16(1, 0);
            if (s == 1) {
                a3_tile(0, 16, 1, 0);
                a3_tile(0, 16, 2, 1);
                a3_tile(0, 16, 2, 2);
            }
        } else {
            const int h = wv - 2;
            if (s >= 1) { // the rank-16 updates by sub-panel s - 1 that wave 0 left behind, each on the wave that applies the same
                          // tile's look-ahead update below (no two waves on one tile)
                if (s == 1) a3_tile(0, 16, h == 0 ? 2 : 1, h == 0 ? 0 : 1); // (3,1) | (2,2); wave 1 has the other three
                if (s == 2) a3_tile(16, 16, 1, h);                          // (3,2) | (3,3)
            }
            if (flags) {
                // The rest of the diagonal block's look-ahead update, at most two tiles per wave and sub-panel (a tile costs
                // ~2.4 k cycles, a sub-panel ~5 k).  The update is additive, so a tile only has to have it before it is next
                // READ: column 0 by A2(0), (1,1) by A1(1), column 1 by A2(1), (2,2) by A1(2), (3,2) by A2(2), (3,3) by A1(3).
                // (A tile that also receives a deferred rank-16 update in the same phase stays on the wave that applies that.)
                if (Wprev) {
                    if (s == 0) { if (h == 0) { tile16(2, 0); tile16(3, 0); } else { tile16(1, 1); tile16(2, 1); } }
                    if (s == 1) { if (h == 0) tile16(3, 1); else tile16(2, 2); }
                    if (s == 2) { if (h == 0) tile16(3, 2); else tile16(3, 3); }
                }
            } else {
                if (Wprev && s == 0) { // the rest of the diagonal block's look-ahead update, hidden under A1(0)
                    if (h == 0) { tile16(2, 0); tile16(3, 0); tile16(1, 1); tile16(2, 1); }
                    else { tile16(3, 1); tile16(2, 2); tile16(3, 2); tile16(3, 3); }
                }
                if (own_rows && (s == 1 || s == 2)) // look-ahead update of the rows below, under A1(1) and A1(2), out of line
                    ba_update_quad_call<T, NB>(ld, p0 - NB, rown, p0, S, Wprev, 2 * (s - 1) + h);
            }
            // rows 1 and 2 of W (the last row with pivots follows the loop): under A1(3); a block with at most 48 pivots has
            // no fourth sub-panel, its row 1 is built under A1(2)
            if (s == 2 && h == 0 && nb <= 48) w_tile(1, 0, 0);
            if (flags && own_rows && s == 3 && h == 1 && lane == 0) {
                // the look-ahead update of this workgroup's rows by another workgroup of this launch (see k_ldlt_step) was
                // finished long ago; a wave with time to spare makes sure here, so that the L2 round trip of the check is not
                // in front of the row GEMM -- every wave reads the rows behind the barriers that follow
                int spins = 0;
                const int spin_limit = fault ? (1 << 8) : BA_FLAG_SPINS;
                while (__hip_atomic_load(&flags[rown / NB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && ++spins < spin_limit)
                    __builtin_amdgcn_s_sleep(4);
                // pairs with the release store of the row workgroup (k_ldlt_step): its rows are visible behind this fence
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // A wait that ran out means the row GEMM below would read rows without their update: the factor would be finite
                // and wrong.  Loud instead: the error word travels to the host with the scalars of the trial (BA_ERR_HIP).
                if (spins >= spin_limit && errw) __hip_atomic_store(errw, (T)BA_DEVERR_ROW_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (s == 3) {
                // ... and the sums of row 3, which do not need W_33 (being built by wave 1 right now): only the three
                // 16 x 16 products W_3c = -W_33 T_c remain behind the loop
                if (h == 0) { w_tile(1, 0, 0); w_tile(2, 0, 0); w_sum(3, 0, 0); }
                else { w_tile(2, 1, 1); w_sum(3, 1, 1); w_sum(3, 2, 2); }
            }
        }
        BA_STAMP_OWN(6);
        __syncthreads();
        BA_STAMP_SEG(0);
        if (s == 3 || c0 + 16 >= NB) break;
        // ---- A2: tiles below, Y^T = W_ss X^T; wave w takes tile t = s + 1 + w
        {
            const int t = s + 1 + wv;
            if (t < 4) {
                // (all LDS operands are requested before the first MFMA / the first store: left to itself the compiler reads
                // each one right in front of its use and pays the LDS round trip eight times in a row)
                acc_t acc;
                T wa[4], xb[4], dv[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    wa[kk] = Wl[c0 + li][c0 + 4 * kk + lk];     // A[j][k] = W_ss[j][k]
                    xb[kk] = Ad[c0 + 4 * kk + lk][16 * t + li]; // B[k][n] = X[n][k]
                }
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    acc[v] = 0;
                    dv[v] = dinv[c0 + ba_crow<T>(lk, v)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc = ba_mfma(wa[kk], xb[kk], acc);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int j = ba_crow<T>(lk, v);
                    const bool piv = j < np; // columns past the last pivot do not take part
                    Ys[j][16 * t + li] = piv ? acc[v] : (T)0;
                    if (piv) Ad[c0 + j][16 * t + li] = acc[v] * dv[v]; // L = Y D^-1
                }
            }
        }
        __syncthreads();
        BA_STAMP_SEG(1);
        BA_STAMP_SEG(2);
        // ---- A3: rank-16 update of the tiles (ti >= tj > s).  Only the next diagonal tile is needed at once: wave 0 does it and
        // goes straight on to the next pivot loop (no barrier); waves 2 and 3 do the others at the start of that loop's phase
        // (they are read by A2 / A1 of later sub-panels, i.e. behind the barrier that ends it).
        if (wv == 0) {
            a3_tile(c0, np, 0, 0);
            ba_wave_lds_order();
        }
        BA_STAMP_SEG(3);
    }
    __syncthreads();
    BA_STAMP_GET(st_t0);
    // ---- the last row of W that has pivots (rows before it were built under A1 of the following sub-panels)
    {
        const int tl = (nb - 1) / 16; // last tile row with pivots
        if (tl == 3) { if (wv < 3) w_fin(3, wv, wv); } // its sums were formed under A1(3) (Ts is not touched in between)
        else if (tl >= 1 && wv < tl) w_tile(tl, wv, wv);
    }
    __syncthreads();
    BA_STAMP_SEG(4);
    // publish the factored block and W (rows >= nb of W are not part of the inverse)
    for (int idx = tid + 256 * blk; idx < NB * NB; idx += 256 * nblk_panel) { // every workgroup writes a slice
        const int r = idx % NB, c = idx / NB;
        // The factored diagonal block itself is only read again when it contains the rhs row (last block column,
        // single workgroup); other workgroups of a wider grid may still be loading the original block from S.
        if (nblk_panel == 1 && c <= r && c < nb) __hip_atomic_store(&S[(size_t)(p0 + c) * ld + p0 + r], Ad[c][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&Winv[c * NB + r], (r <= c && c < nb) ? Wl[c][r] : (T)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (c, r) = (row, column) of W
    }
    // ---- rows below the diagonal block: Y^T = W X^T on the matrix cores.  Wave w owns 16 rows and all four 16-wide column
    // tiles of the panel (40 MFMAs: W is lower triangular); with two workgroups per row block 16 rows and two column tiles
    // (0 and 3 or 1 and 2: 20 MFMAs either way).
    const int r0 = HALVES ? p0 + NB + 64 * rblk + 32 * (blk & 1) + 16 * (wv & 1) : p0 + NB + 64 * blk + 16 * wv;
    if (r0 >= nrows || nb < NB) return;
    const T *const px = S + (size_t)(p0 + lk) * ld + r0 + li; // X[n][k] at px[k / 4 * 4 ld]
    if constexpr (!HALVES) {
        acc_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[t][v] = 0;
        T xall[NB / 4]; // all sixteen loads in flight before the first MFMA (one L2 round trip instead of four)
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) // B[k][n] = X[n][k]; agent-scope load = sc1, served by L2: this CU's L1 may hold
                                            // the pre-update lines
            xall[kk] = __hip_atomic_load(px + kk * (4 * (size_t)ld), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) {
            const T xb = xall[kk];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (kk <= 4 * t + 3) { // W is lower triangular: W[j][k] = 0 for k > j
                    const T wa = Wl[16 * t + li][4 * kk + lk]; // A[j][k]
                    acc[t] = ba_mfma(wa, xb, acc[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int j = 16 * t + ba_crow<T>(lk, v); // column of the panel
                const T yv = acc[t][v];
                // (write-through stores, like every bulk store of the factorisation: what a launch writes should leave the L2s
                // while it runs, not in the release at its end -- the next launch reads it from other XCDs anyway)
                __hip_atomic_store(&Wp[(size_t)j * ld + r0 + li], yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&S[(size_t)(p0 + j) * ld + r0 + li], yv * dinv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    } else {
        T xall[NB / 4];
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++)
            xall[kk] = __hip_atomic_load(px + kk * (4 * (size_t)ld), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_sched_barrier(0);
        auto rowgemm = [&](auto tiles) { // tiles: the two column tiles of this wave (compile-time list)
            constexpr int NT = decltype(tiles)::n;
            acc_t acc[NT];
#pragma unroll
            for (int u = 0; u < NT; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[u][v] = 0;
#pragma unroll
            for (int kk = 0; kk < NB / 4; kk++) {
                const T xb = xall[kk];
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    const int t = decltype(tiles)::t(u);
                    if (kk <= 4 * t + 3) {
                        const T wa = Wl[16 * t + li][4 * kk + lk];
                        acc[u] = ba_mfma(wa, xb, acc[u]);
                    }
                }
            }
            // Waves w and w ^ 2 share their 16 rows: each reads ALL of X there and overwrites ITS two column tiles with L, in place.
            // Every wave's X must have arrived (its MFMAs are done) before any wave stores.  (Found with two workgroups per CU,
            // where the waves of a workgroup drift apart by microseconds: tile 3 of a row block was computed from tiles 1 and 2 of
            // L instead of X.)
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NT; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int j = 16 * decltype(tiles)::t(u) + ba_crow<T>(lk, v);
                    const T yv = acc[u][v];
                    __hip_atomic_store(&Wp[(size_t)j * ld + r0 + li], yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S[(size_t)(p0 + j) * ld + r0 + li], yv * dinv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        };
        if (wv < 2) rowgemm(ba_tiles<2, 0, 3, 0, 0>());
        else rowgemm(ba_tiles<2, 1, 2, 0, 0>());
    }
    BA_STAMP_SEG(5);
    BA_STAMP_FLUSH
}

#undef BA_TS

// Stand-alone panel step (first block column, dense bench).
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_panel(int nrows, int ncols, int ld, int p0, T *__restrict__ S, T *__restrict__ Wp,
                                                    T *__restrict__ Winv, int *__restrict__ flags = nullptr, int nflags = 0)
{
    if (flags && blockIdx.x == 0) // hand-off flags of the fused steps that follow (k_ldlt_step<INL = true>): cleared per factorisation
        for (int i = threadIdx.x; i < nflags; i += 256) flags[i] = 0;
    __shared__ T Ad[NB][NB + 1], Wl[NB][NB + 1];
    ba_panel_body<T, NB, false>(nrows, ncols, ld, p0, S, Wp, Winv, nullptr, blockIdx.x, gridDim.x, Ad, Wl);
}

// Fused step with look-ahead: ONE launch per block column p0 >= 64.
//   workgroups [0, npanel)        : panel step of block column p0 (after applying the previous panel's update to it);
//   workgroups [npanel, gridDim.x): the rest of the trailing update of block column p0 - 64 (tiles with columns >= p0 + 64).
// The two groups touch disjoint parts of S; Wp is double-buffered (Wprev read, Wp written).  The panel's 2313-long pivot
// recurrence is the critical path of the factorisation; this hides the MFMA update behind it.
//
// INL = true (D <~ 3000: one workgroup per CU, the panel is the critical path): nq further workgroups in FRONT of the grid take
// the look-ahead update of the panel workgroups' rows (one 64 x 64 tile of block column p0 each, written with agent-scope
// stores), announce it in flags[row block] and leave; panel workgroup b picks its rows up just before its row GEMM.  They are
// dispatched before the workgroups that wait for them, so the wait cannot starve them.  INL = false keeps that update
// inside the panel workgroup (two helper waves, out of line), nq = 0.
// Macro-tile part of a launch (k_ldlt_step2): panels at columns pM and pM + 64 (Y in W1, W2) applied to the 128 x 128 tiles
// rows base + 128 mi, columns base + 128 mj, mj <= mi (everything behind the block column that the launch itself factors).
template <typename T> struct ba_macro_job { const T *W1, *W2; int pM, base, count; }; // base: first row / column of the tiles

template <typename T, int NB, bool INL, bool MACRO>
__device__ __forceinline__ void ba_step_body(int nrows, int ncols, int ld, int p0, int npanel, T *__restrict__ S, T *__restrict__ Wp,
                                             const T *__restrict__ Wprev, T *__restrict__ Winv, int nq, int *__restrict__ flags,
                                             T *__restrict__ errw, int upd_mode, int n64, const ba_macro_job<T> &mj, int fault)
{
    __shared__ T Ad[NB][NB + 1], Wl[NB][NB + 1];
    int bid = blockIdx.x;
    if (INL && bid < nq) {
        const int rown = p0 + NB + 64 * bid;
        ba_update_quad<T, NB, false, true>(ld, p0 - NB, rown, p0, false, S, Wprev, nullptr, threadIdx.x >> 6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through stores have left
        __syncthreads();
        // release: the tile's (write-through) stores of every wave, ordered by the barrier above, are visible at agent scope before the flag
        if (threadIdx.x == 0 && !fault) __hip_atomic_store(&flags[rown / NB], p0 / NB, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (INL) bid -= nq;
    if (bid < npanel) {
        ba_panel_body<T, NB, INL>(nrows, ncols, ld, p0, S, Wp, Winv, Wprev, bid, npanel, Ad, Wl, INL ? flags : nullptr, p0 / NB, errw, fault);
        return;
    }
    // Trailing update by the workgroups behind the panel workgroups.  First n64 workgroups with 64 x 64 tiles and panel p - 1 alone:
    // upd_mode 0 on every tile of the set {(ti, tj): 1 <= tj <= ti, tj < ntc} (rows p0 + 64 ti, columns p0 + 64 tj), upd_mode 1 on
    // the NEXT block column only (tj = 1: the one the next step factors).  Behind them (large matrices, k_ldlt_step2) the macro tiles
    // of a PAIR of earlier panels: the trailing matrix is read and written once per two panels (K = 128).
    const int ntc = (ncols - p0 + 63) / 64;
    int u = bid - npanel, ti = 1, tj;
    if (MACRO && u >= n64) {
        u -= n64;
        if (u >= mj.count) return;
        // (tile rows in order, a row's tiles next to each other: consecutive workgroups = the eight XCDs share a row's Y operand.  Round 4
        // tried one XCD per tile row -- a short row paired with a long one per XCD, operands served by that XCD's L2 alone: 8.19 against
        // 7.78 ms at D = 9216, profiles/EXPERIMENTS.md 2)
        const int base = mj.base, nmc = ((ncols - base + 63) / 64 + 1) / 2;
        int mi = 0;
        for (;; mi++) {
            const int cnt = min(mi + 1, nmc);
            if (u < cnt) break;
            u -= cnt;
        }
        const int row0 = base + 128 * mi, col0 = base + 128 * u;
        if (row0 >= nrows || col0 >= ncols) return;
        ba_update_macro<T, NB>(nrows, ncols, ld, mj.pM, row0, col0, S, mj.W1, mj.W2, &Ad[0][0], &Wl[0][0]);
        return;
    }
    if (upd_mode == 1) { ti = 1 + u; tj = 1; }
    else {
        for (;; ti++) {
            const int cnt = min(ti, ntc - 1);
            if (u < cnt) break;
            u -= cnt;
        }
        tj = 1 + u;
    }
    const int row0 = p0 + 64 * ti, col0 = p0 + 64 * tj;
    if (row0 >= nrows || col0 >= ncols) return;
    // (write-through stores: the 17 MB a launch writes leave the L2s while it runs, not in the release at its end)
    ba_update_tile<T, NB, false, true>(ld, p0 - NB, row0, col0, ti == tj, S, Wprev, nullptr);
}

template <typename T, int NB, bool INL>
__global__ __launch_bounds__(256) void k_ldlt_step(int nrows, int ncols, int ld, int p0, int npanel, T *__restrict__ S,
                                                   T *__restrict__ Wp, const T *__restrict__ Wprev, T *__restrict__ Winv,
                                                   int nq = 0, int *__restrict__ flags = nullptr, T *__restrict__ errw = nullptr, int fault = 0)
{
    static_assert(INL, "the two-per-CU variant is k_ldlt_step2");
    ba_step_body<T, NB, true, false>(nrows, ncols, ld, p0, npanel, S, Wp, Wprev, Winv, nq, flags, errw, 0, 1 << 30, ba_macro_job<T>{}, fault);
}

// The two-workgroups-per-CU variant for update-bound sizes (no dynamic-LDS request): the same panel structure -- row workgroups in
// front, two panel workgroups per row block -- and behind them the 64 x 64 tiles of one panel (upd_mode 0 / 1) and the macro tiles of
// a pair of panels.  Built with -mllvm -amdgpu-mfma-vgpr-form (Makefile): with the accumulators of the macro-tile update in AGPRs
// the kernel would need the panel's VGPRs PLUS 128 AGPRs (the unified file is split, not shared) and lose the second workgroup per
// CU; tests/test_kernel_resources.py pins <= 256.
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_step2(int nrows, int ncols, int ld, int p0, int npanel, T *__restrict__ S, T *__restrict__ Wp,
                                                    const T *__restrict__ Wprev, T *__restrict__ Winv, int nq, int *__restrict__ flags,
                                                    T *__restrict__ errw, int upd_mode, int n64, ba_macro_job<T> mj, int fault = 0)
{
    ba_step_body<T, NB, true, true>(nrows, ncols, ld, p0, npanel, S, Wp, Wprev, Winv, nq, flags, errw, upd_mode, n64, mj, fault);
}

// Trailing update of one 64 x 64 tile with the 64-wide panel at block column p0: C_ij -= sum_k Y_ik L_jk.
// Wave w owns a 32 x 32 quadrant (2 x 2 accumulators: two A and two B fragments feed four MFMAs).  The MFMA computes
// the TRANSPOSED tile  C^T[j][i] -= sum_k L[j][k] Y[i][k]  so that the 16-wide "column" index of the C/D fragment runs
// along the rows of S (contiguous in the column-major matrix): every accumulator load / store instruction touches
// 4 columns x 128 contiguous bytes.  Operands come straight from L2 (the panel is a few MB at most); 8 k-steps are in
// flight at once to cover the L2 latency.  LOWER: skip the strictly upper quadrant (diagonal tiles).
// TOLDS: the C tile lives in the LDS image Cl[col][row] (64 x 65) instead of S (diagonal block inside the panel step).
// STSC: the results leave with agent-scope (sc1, write-through) stores: another workgroup of the SAME launch reads them.
template <typename T, int NB, bool TOLDS, bool STSC>
__device__ __forceinline__ void ba_update_quad(int ld, int p0, int row0t, int col0t, bool lower, T *__restrict__ S,
                                               const T *__restrict__ Wp, T (*Cl)[NB + 1], int quad)
{
    const int lane = threadIdx.x & 63;
    const int qr = 32 * (quad >> 1), qc = 32 * (quad & 1);
    if (lower && qc > qr) return;
    const int row0 = row0t + qr, col0 = col0t + qc;
    const int li = lane & 15, lk = lane >> 4;
    typename ba_acc<T>::type acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int cc = 16 * t + ba_crow<T>(lk, v), rr = 16 * u + li;
                acc[t][u][v] = TOLDS ? Cl[qc + cc][qr + rr] : S[(size_t)(col0 + cc) * ld + row0 + rr];
            }
    // k-steps whose operands are in flight together: all sixteen for the diagonal block (latency: it is on the critical
    // path), eight for the trailing tiles (throughput: registers)
    constexpr int CH = TOLDS ? NB / 4 : ((NB / 4 < 8) ? NB / 4 : 8);
#pragma unroll
    for (int half = 0; half < (NB / 4) / CH; half++) {
        T a[CH][2], b[CH][2];
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int kk = CH * half + q;
#pragma unroll
            for (int t = 0; t < 2; t++) a[q][t] = S[(size_t)(p0 + 4 * kk + lk) * ld + col0 + 16 * t + li]; // A[j][k] = L[j][k] (negated at use)
#pragma unroll
            for (int u = 0; u < 2; u++) b[q][u] = Wp[(size_t)(4 * kk + lk) * ld + row0 + 16 * u + li];        // B[k][i] = Y[i][k]
        }
#pragma unroll
        for (int q = 0; q < CH; q++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < 2; u++) acc[t][u] = ba_mfma(-a[q][t], b[q][u], acc[t][u]);
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int cc = 16 * t + ba_crow<T>(lk, v), rr = 16 * u + li;
                if (TOLDS) Cl[qc + cc][qr + rr] = acc[t][u][v];
                else if (STSC) __hip_atomic_store(&S[(size_t)(col0 + cc) * ld + row0 + rr], acc[t][u][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else S[(size_t)(col0 + cc) * ld + row0 + rr] = acc[t][u][v];
            }
}

template <typename T, int NB, bool TOLDS, bool STSC>
__device__ __forceinline__ void ba_update_tile(int ld, int p0, int row0t, int col0t, bool lower, T *__restrict__ S,
                                               const T *__restrict__ Wp, T (*Cl)[NB + 1])
{
    ba_update_quad<T, NB, TOLDS, STSC>(ld, p0, row0t, col0t, lower, S, Wp, Cl, threadIdx.x >> 6); // wave w owns quadrant w
}

// Trailing update of one 128 x 128 MACRO tile with TWO panels (block columns pA and pA + 64: K = 128), operands staged through
// LDS -- the update of a large matrix (config 5).  The quadrant update above feeds every MFMA from one 8-byte L2 load per
// lane; at D = 9216 that path, not the matrix cores, sets the pace (35 % of the fp64 rate).  Here a workgroup owns 128 rows x
// 128 columns of C (wave w the 64 x 64 quadrant (w >> 1, w & 1): 4 x 4 accumulator tiles, 128 registers), and the operands
// L[col0 + j][k] and Y[row0 + i][k] pass through LDS in stages of 16 values of k: per stage 2 x 16 KiB arrive with 16-byte
// loads (one 1 KiB row of the column-major panel per wave and instruction), requested one stage ahead of the MFMAs that use them
// and parked in the other half of the two LDS images the panel workgroups of the same launch use for the diagonal block (Ad, Wl:
// 2 x 32.5 KiB), so the launch keeps its two workgroups per CU.  4x fewer operand bytes leave the L2 than with the quadrant
// update, and a fragment is a conflict-free ds_read_b64: stage layout [k][row ^ 16 (k & 1)] -- the two values of k that one
// half-wave reads land in different halves of the banks.
// Waves whose quadrant lies outside the matrix or strictly above the diagonal skip their MFMAs and stores (they still help with
// the staging); operand rows past the matrix are clamped (they feed skipped quadrants only).
template <typename T> struct ba_vec2;
template <> struct ba_vec2<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct ba_vec2<float> { typedef float type __attribute__((ext_vector_type(2))); };

template <typename T, int NB>
__device__ __forceinline__ void ba_update_macro(int nrows, int ncols, int ld, int pA, int row0, int col0, T *__restrict__ S,
                                                const T *__restrict__ W1, const T *__restrict__ W2, T *__restrict__ As, T *__restrict__ Bs)
{
    typedef typename ba_vec2<T>::type v2;
    constexpr int KS = 16, MT = 128, NST = 2 * NB / KS; // k per stage, macro tile, stages
    static_assert(2 * KS * MT <= NB * (NB + 1), "two stages of one operand fit one LDS image");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wr = 64 * (wv >> 1), wc = 64 * (wv & 1);
    const bool live = row0 + wr < nrows && col0 + wc < ncols && col0 + wc <= row0 + wr;
    // staging: this thread moves, per stage and operand, 4 pairs: k = kq + 4 it (kq = tid >> 6), rows 2 (tid & 63), +1
    const int sj = 2 * (tid & 63), kq = tid >> 6;
    const int ja = min(col0 + sj, ld - 2), ib = min(row0 + sj, ld - 2);
    v2 ga[4], gb[4];
    auto request = [&](int st) {
        const int panel = st / (NB / KS), k0 = (st % (NB / KS)) * KS; // first / second panel, k offset inside it
        const T *const Lp = S + (size_t)(pA + NB * panel + k0 + kq) * ld + ja;
        const T *const Yp = (panel ? W2 : W1) + (size_t)(k0 + kq) * ld + ib;
#pragma unroll
        for (int it = 0; it < 4; it++) {
#ifdef BA_KO_MACRO_STAGE /* knock-out experiment (scripts/bench_dense.hip): no operand loads */
            ga[it] = v2{(T)st, (T)it}; gb[it] = v2{(T)it, (T)st}; (void)Lp; (void)Yp;
#else
            ga[it] = *(const v2 *)(Lp + (size_t)(4 * it) * ld);
            gb[it] = *(const v2 *)(Yp + (size_t)(4 * it) * ld);
#endif
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int k = kq + 4 * it; // (k & 1) == (kq & 1)
            const int o = buf * KS * MT + k * MT + (sj ^ (16 * (kq & 1)));
            *(v2 *)(As + o) = ga[it];
            *(v2 *)(Bs + o) = gb[it];
        }
    };
    request(0);
    typename ba_acc<T>::type acc[4][4];
    if (live) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++)
#ifdef BA_KO_MACRO_C /* knock-out experiment: the C tile is neither read nor written */
                    acc[t][u][v] = (T)(t + u + v);
#else
                    acc[t][u][v] = S[(size_t)(col0 + wc + 16 * t + ba_crow<T>(lk, v)) * ld + row0 + wr + 16 * u + li];
#endif
    }
    park(0);
    __syncthreads();
    for (int st = 0; st < NST; st++) {
        if (st + 1 < NST) request(st + 1);
        if (live) {
            const T *const Ab = As + (st & 1) * KS * MT, *const Bb = Bs + (st & 1) * KS * MT;
#pragma unroll
            for (int q = 0; q < KS / 4; q++) {
                const int k = 4 * q + lk, sw = 16 * (lk & 1);
                T a[4], b[4];
#pragma unroll
                for (int t = 0; t < 4; t++) a[t] = Ab[k * MT + ((wc + 16 * t + li) ^ sw)];
#pragma unroll
                for (int u = 0; u < 4; u++) b[u] = Bb[k * MT + ((wr + 16 * u + li) ^ sw)];
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
#ifdef BA_KO_MACRO_MFMA /* knock-out experiment: no matrix instructions */
                    for (int u = 0; u < 4; u++) acc[t][u][(t + u) & 3] += a[t] * b[u];
#else
                    for (int u = 0; u < 4; u++) acc[t][u] = ba_mfma(-a[t], b[u], acc[t][u]);
#endif
            }
        }
        if (st + 1 < NST) park((st + 1) & 1);
        __syncthreads();
    }
#ifdef BA_KO_MACRO_C
    if (live) {
        T sum = 0;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) sum += acc[t][u][v];
        if (sum == (T)123.456) S[0] = sum; // (keeps the accumulators alive)
    }
#else
    if (live) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++)
                    __hip_atomic_store(&S[(size_t)(col0 + wc + 16 * t + ba_crow<T>(lk, v)) * ld + row0 + wr + 16 * u + li], acc[t][u][v],
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

// The same macro-tile update as a kernel of its own with EIGHT waves per 128 x 128 tile (wave w: rows 64 (w >> 2), columns 32 (w & 3):
// 2 x 4 accumulator tiles, 64 registers): two workgroups per CU are then four waves per SIMD instead of two -- twice as many
// instruction streams whose C-tile prologue / epilogue and operand waits can hide under somebody's MFMAs (round 4's knock-outs:
// 27.9 GB of traffic and 2.9 ms of matrix work per factorisation at D = 9216 overlap by ~1 ms in the fused form).
// Experiment switch BA_LDLT_MACRO8=1 (ba_ldlt_factor); grid = the macro tiles of ba_macro_job, in the same order.
template <typename T, int NB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ldlt_macro8(int nrows, int ncols, int ld, ba_macro_job<T> mj, T *__restrict__ S)
{
    typedef typename ba_vec2<T>::type v2;
    constexpr int KS = 16, MT = 128, NST = 2 * NB / KS;
    __shared__ __attribute__((aligned(16))) T As[2 * KS * MT], Bs[2 * KS * MT];
    int u = blockIdx.x;
    if (u >= mj.count) return;
    const int base = mj.base, nmc = ((ncols - base + 63) / 64 + 1) / 2;
    int mi = 0;
    for (;; mi++) {
        const int cnt = min(mi + 1, nmc);
        if (u < cnt) break;
        u -= cnt;
    }
    const int row0 = base + 128 * mi, col0 = base + 128 * u;
    if (row0 >= nrows || col0 >= ncols) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wr = 64 * (wv >> 2), wc = 32 * (wv & 3);
    const bool live = row0 + wr < nrows && col0 + wc < ncols && col0 + wc < row0 + wr + 64;
    const int sj = 2 * (tid & 63), kq = tid >> 6; // staging: per stage and operand 2 pairs, k = kq + 8 it
    const int ja = min(col0 + sj, ld - 2), ib = min(row0 + sj, ld - 2);
    const int pA = mj.pM;
    v2 ga[2], gb[2];
    auto request = [&](int st) {
        const int panel = st / (NB / KS), k0 = (st % (NB / KS)) * KS;
        const T *const Lp = S + (size_t)(pA + NB * panel + k0 + kq) * ld + ja;
        const T *const Yp = (panel ? mj.W2 : mj.W1) + (size_t)(k0 + kq) * ld + ib;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            ga[it] = *(const v2 *)(Lp + (size_t)(8 * it) * ld);
            gb[it] = *(const v2 *)(Yp + (size_t)(8 * it) * ld);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int k = kq + 8 * it; // (k & 1) == (kq & 1)
            const int o = buf * KS * MT + k * MT + (sj ^ (16 * (kq & 1)));
            *(v2 *)(As + o) = ga[it];
            *(v2 *)(Bs + o) = gb[it];
        }
    };
    request(0);
    typename ba_acc<T>::type acc[2][4];
    if (live) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int uu = 0; uu < 4; uu++)
#pragma unroll
                for (int v = 0; v < 4; v++)
                    acc[t][uu][v] = S[(size_t)(col0 + wc + 16 * t + ba_crow<T>(lk, v)) * ld + row0 + wr + 16 * uu + li];
    }
    park(0);
    __syncthreads();
    for (int st = 0; st < NST; st++) {
        if (st + 1 < NST) request(st + 1);
        if (live) {
            const T *const Ab = As + (st & 1) * KS * MT, *const Bb = Bs + (st & 1) * KS * MT;
#pragma unroll
            for (int q = 0; q < KS / 4; q++) {
                const int k = 4 * q + lk, sw = 16 * (lk & 1);
                T a[2], b[4];
#pragma unroll
                for (int t = 0; t < 2; t++) a[t] = Ab[k * MT + ((wc + 16 * t + li) ^ sw)];
#pragma unroll
                for (int uu = 0; uu < 4; uu++) b[uu] = Bb[k * MT + ((wr + 16 * uu + li) ^ sw)];
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int uu = 0; uu < 4; uu++) acc[t][uu] = ba_mfma(-a[t], b[uu], acc[t][uu]);
            }
        }
        if (st + 1 < NST) park((st + 1) & 1);
        __syncthreads();
    }
    if (live) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int uu = 0; uu < 4; uu++)
#pragma unroll
                for (int v = 0; v < 4; v++)
                    __hip_atomic_store(&S[(size_t)(col0 + wc + 16 * t + ba_crow<T>(lk, v)) * ld + row0 + wr + 16 * uu + li], acc[t][uu][v],
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Out-of-line copy for the call sites inside the panel's sub-panel loop: inlined there, the update's ~100 live
// registers are merged into the allocation of the fully unrolled pivot loop (251 VGPRs instead of 95).
template <typename T, int NB>
__device__ __attribute__((noinline)) void ba_update_quad_call(int ld, int p0, int row0t, int col0t, T *S, const T *Wp, int quad)
{
    ba_update_quad<T, NB, false>(ld, p0, row0t, col0t, false, S, Wp, nullptr, quad);
}

// Stand-alone trailing update (one launch per block column; kept for the non-fused path and the dense bench).
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_update(int nrows, int ncols, int ld, int p0, T *__restrict__ S, const T *__restrict__ Wp,
                                                     int owners = 1, int owner = 0 /* distributed factor: only the block columns q with q % owners == owner */)
{
    const int p1 = p0 + NB;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    if (owners > 1 && (p1 / NB + tj) % owners != owner) return;
    const int row0 = p1 + 64 * ti, col0 = p1 + 64 * tj;
    if (row0 >= nrows || col0 >= ncols) return;
    ba_update_tile<T, NB, false>(ld, p0, row0, col0, ti == tj, S, Wp, nullptr);
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
    // first pass of the elimination loop below: its column entries do not depend on x, so they are requested up front and
    // arrive under the GEMV (one L2 round trip less per launch; there are nblk launches in a row)
    const int lane = tid & 63, w = tid >> 6, cq = lane & 15, rq = lane >> 4;
    const int cb0 = (blockIdx.x * 4 + w) * 16;
    T cpre[NB / 4];
    if (cb0 + cq < p0) {
        const T *col = S + (size_t)(cb0 + cq) * ld + p0 + (NB / 4) * rq;
#pragma unroll
        for (int t = 0; t < NB / 4; t++) cpre[t] = col[t];
    }
    const T zpre = (cb0 + cq < p0 && rq == 0) ? S[(size_t)(cb0 + cq) * ld + zrow] : (T)0;
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
    // elimination from the earlier unknowns: a wave takes 16 columns per pass, lane (cq, rq) sums 16 of the 64 rows of
    // column cq (128 contiguous bytes), two butterfly steps combine the four row quarters
    T xr[NB / 4];
#pragma unroll
    for (int t = 0; t < NB / 4; t++) xr[t] = xs[(NB / 4) * rq + t];
    for (int cb = cb0; cb < p0; cb += gridDim.x * 64) {
        const int c = cb + cq;
        T a = 0;
        if (c < p0) {
            if (cb == cb0) {
#pragma unroll
                for (int t = 0; t < NB / 4; t++) a += cpre[t] * xr[t];
            } else {
                const T *col = S + (size_t)c * ld + p0 + (NB / 4) * rq;
#pragma unroll
                for (int t = 0; t < NB / 4; t++) a += col[t] * xr[t];
            }
        }
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (rq == 0 && c < p0) S[(size_t)c * ld + zrow] = ((cb == cb0) ? zpre : S[(size_t)c * ld + zrow]) - a;
    }
}

// Backward sweep, two block columns per launch (blocks pb and pb + 1; the per-launch cost of k_ldlt_backstep is three
// dependent L2 round trips plus the kernel boundary, not arithmetic).  Every operand that does not depend on x -- both
// inverses, the coupling block L(pb + 1, pb), the first pass of the elimination columns -- is requested before the first
// barrier; the launch then has ONE exposed round trip and a chain of LDS phases:
//   x1 = W1^T z1;  z0 -= L10^T x1;  x0 = W0^T z0;  z_c -= sum_r L(r, c) [x0; x1](r)  for the earlier columns c.
template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_backpair(int ncols, int ld, int zrow, int pb, T *__restrict__ S, const T *__restrict__ Winv,
                                                       T *__restrict__ x)
{
    static_assert(NB == 64, "written for 64-wide block columns");
    __shared__ T zs[2][NB], xs[2 * NB], part[4][NB];
    const int tid = threadIdx.x, j = tid & 63, q = tid >> 6;
    const int p0 = pb * NB, p1 = p0 + NB;
    const T *W0 = Winv + (size_t)pb * NB * NB, *W1 = W0 + (size_t)NB * NB;
    T w1[16], w0[16], l10[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int k = 16 * q + t;
        w1[t] = W1[k * NB + j];                       // W1[k][j]
        w0[t] = W0[k * NB + j];
        l10[t] = S[(size_t)(p0 + j) * ld + p1 + k];   // L(p1 + k, p0 + j)
    }
    // elimination columns of the first pass: lane (cq, rq) sums the rq-th quarter of the 128 rows of column cq
    const int lane = tid & 63, w = tid >> 6, cq = lane & 15, rq = lane >> 4;
    const int cb0 = (blockIdx.x * 4 + w) * 16;
    T cpre[32];
    const bool have0 = cb0 + cq < p0;
    if (have0) {
        const T *col = S + (size_t)(cb0 + cq) * ld + p0 + 32 * rq;
#pragma unroll
        for (int t = 0; t < 32; t++) cpre[t] = col[t];
    }
    const T zpre = (have0 && rq == 0) ? S[(size_t)(cb0 + cq) * ld + zrow] : (T)0;
    if (tid < 2 * NB) {
        const int c = p0 + tid;
        zs[tid >> 6][tid & 63] = (c < ncols) ? S[(size_t)c * ld + zrow] : (T)0;
    }
    __syncthreads();
    T a = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) a += w1[t] * zs[1][16 * q + t];
    part[q][j] = a;
    __syncthreads();
    if (tid < NB) xs[NB + tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    __syncthreads();
    a = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) a += l10[t] * xs[NB + 16 * q + t];
    part[q][j] = a;
    __syncthreads();
    if (tid < NB) zs[0][tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    __syncthreads();
    a = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) a += w0[t] * zs[0][16 * q + t];
    part[q][j] = a;
    __syncthreads();
    if (tid < NB) xs[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    __syncthreads();
    if (blockIdx.x == 0 && tid < 2 * NB && p0 + tid < ncols) x[p0 + tid] = xs[tid];
    T xr[32];
#pragma unroll
    for (int t = 0; t < 32; t++) xr[t] = xs[32 * rq + t];
    for (int cb = cb0; cb < p0; cb += gridDim.x * 64) {
        const int c = cb + cq;
        T s_ = 0;
        if (c < p0) {
            if (cb == cb0) {
#pragma unroll
                for (int t = 0; t < 32; t++) s_ += cpre[t] * xr[t];
            } else {
                const T *col = S + (size_t)c * ld + p0 + 32 * rq;
#pragma unroll
                for (int t = 0; t < 32; t++) s_ += col[t] * xr[t];
            }
        }
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        if (rq == 0 && c < p0) S[(size_t)c * ld + zrow] = ((cb == cb0) ? zpre : S[(size_t)c * ld + zrow]) - s_;
    }
}

// Backward sweep as ONE launch (data flow between workgroups instead of a launch per step: a back-sweep launch is ~3.5 us of
// kernel boundary around ~3 us of work).  Workgroup g owns a group of two block columns (the first group is a single
// block column when their number is odd).  It eliminates the later groups from its right-hand side as their unknowns
// appear -- each x(i) is an 8-byte granule that its owner publishes with an agent-scope (write-through) store over a
// sentinel, so a consumer polls the very datum it needs and no flag or ordering is involved -- with the 128 x 128 block of
// L for the next group already in registers; then it solves its own group like k_ldlt_backpair and publishes.  The chain is
// one hop per group: poll, 64 FMAs per thread, three LDS GEMV phases.  A group only waits for groups dispatched BEFORE it (the last
// group has the lowest workgroup index) IF workgroups are dispatched in index order -- observed, not promised by HIP: the host caps the
// grid at one workgroup per CU (ba_ldlt_backsweep: max_groups) so that the whole launch is resident, which is what correctness rests
// on (38 workgroups at config 4, 144 at config 5; the spins are bounded anyway).  x must hold the sentinel on entry (k_post_reduce / k_fill_sentinel).
template <typename T>
__global__ void k_fill_sentinel(int n, T *__restrict__ x)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = ba_sentinel<T>();
}

template <typename T, int NB>
__global__ __launch_bounds__(256) void k_ldlt_backflow(int ncols, int ld, int zrow, int nblk, T *__restrict__ S, const T *__restrict__ Winv,
                                                       T *__restrict__ x, T *__restrict__ zh /* [columns], armed like x */, T *__restrict__ errw = nullptr,
                                                       int spin_limit = BA_SWEEP_SPINS, int skip_group = -1 /* self-test only: this group never publishes */)
{
    static_assert(NB == 64, "written for 64-wide block columns");
    __shared__ T zs[2 * NB], xin[2 * NB], xs[2 * NB], part[4][NB], part2[2][2 * NB];
    const int wg = blockIdx.x, nwg = gridDim.x;
    // TWO workgroups per group.  A workgroup pulls the 128 KB block of L of every later group through one CU (~50 GB/s: 2.6 us per
    // block), and a hop of the chain could not be shorter than that.  The later groups are dealt to the two by the parity of their
    // distance: the main workgroup (role 0) takes g + 1, g + 3, ... -- the last one to arrive among them -- and solves the group;
    // the helper (role 1) takes g + 2, g + 4, ..., so it is done one hop BEFORE the chain reaches the group and hands its partial
    // sums over (zh, same sentinel protocol as x) off the critical path.  Each now needs a block every second hop.
    // The LAST group goes first in dispatch order: a group waits for the groups behind it only, so every wait is for a workgroup that
    // was dispatched earlier IF dispatch follows the index order (an assumption; the grid is capped to be fully resident regardless).
    // (and a group's helper in front of its main workgroup, which waits for the helper's partial sums)
    const int tid = threadIdx.x, G = nwg >> 1, g = G - 1 - (wg >> 1), role = (wg & 1) ^ 1;
    if (g == skip_group) return;
    const bool odd = (nblk & 1) != 0;
    const int fb = odd ? max(0, 2 * g - 1) : 2 * g, nbg = (odd && g == 0) ? 1 : 2; // first block column / block columns of this group
    const int c0 = fb * NB, ncg = nbg * NB;
    // operands of the group's own solve: thread (j, q) holds 16 rows of column j of W1, W0 and L10
    const int j = tid & 63, q = tid >> 6;
    const T *W0 = Winv + (size_t)fb * NB * NB, *W1 = W0 + (size_t)NB * NB;
    // (unconditional loads from addresses that are valid either way: a load under `nbg == 2 ? ... : 0` is compiled as a branch
    // with a wait behind every single load -- sixteen L2 round trips in a row at the head of the sweep)
    const T *W1p = nbg == 2 ? W1 : W0;
    T w1[16], w0[16], l10[16];
    if (role == 0) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int k = 16 * q + t;
            w0[t] = W0[k * NB + j];
            w1[t] = W1p[k * NB + j];
            l10[t] = S[(size_t)(c0 + j) * ld + c0 + NB + k]; // L(c0 + 64 + k, c0 + j); unused (finite padding) when nbg == 1
        }
    }
    const int zc = c0 + (tid < ncg ? tid : 0);
    T zv = S[(size_t)(zc < ncols ? zc : c0) * ld + zrow];
    if (tid < 2 * NB) zs[tid] = (role == 0 && tid < ncg && c0 + tid < ncols) ? zv : (T)0; // (the helper starts from zero)
    // elimination of the later groups: thread (j2, h) sums half h of the 128 rows of column j2
    const int j2 = tid & 127, h = tid >> 7;
    T lpre[NB];
    auto load_L = [&](int qg) {
        const int r0 = (odd ? 2 * qg - 1 : 2 * qg) * NB + NB * h; // (qg >= 1: two block columns)
        const T *col = S + (size_t)(c0 + (j2 < ncg ? j2 : 0)) * ld + r0;
#pragma unroll
        for (int t = 0; t < NB; t++) lpre[t] = col[t];
    };
    // this workgroup's later groups: qg = qtop, qtop - 2, ... > g, with (qg - g) odd for the main one, even for the helper
    const int want = role == 0 ? 1 : 0;
    const int qtop = (((G - 1 - g) & 1) == want) ? G - 1 : G - 2;
    if (qtop > g) load_L(qtop);
    __syncthreads();
    for (int qg = qtop; qg > g; qg -= 2) {
        const int r0 = (odd ? 2 * qg - 1 : 2 * qg) * NB;
        if (tid < 2 * NB) {
            T xv = (T)0;
            if (r0 + tid < ncols) {
                int spins = 0;
                do xv = __hip_atomic_load(&x[r0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (ba_is_sentinel(xv) && ++spins < spin_limit);
                // never published: the sweep would go on with the sentinel (a NaN) -- report it instead of a silently rejected step
                if (ba_is_sentinel(xv) && errw) __hip_atomic_store(errw, (T)BA_DEVERR_SWEEP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            xin[tid] = xv;
        }
        __syncthreads();
        T a = 0;
#pragma unroll
        for (int t = 0; t < NB; t++) a += lpre[t] * xin[NB * h + t];
        part2[h][j2] = a;
        if (qg - 2 > g) load_L(qg - 2); // the next block is in flight while this one is reduced and the next x awaited
        __syncthreads();
        if (tid < 2 * NB) zs[tid] -= part2[0][tid] + part2[1][tid];
        __syncthreads();
    }
    if (role == 1) { // hand the partial sums to the main workgroup and leave
        if (tid < ncg) __hip_atomic_store(&zh[c0 + tid], zs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (tid < ncg) {
        T hv;
        int spins = 0;
        do hv = __hip_atomic_load(&zh[c0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ba_is_sentinel(hv) && ++spins < spin_limit);
        if (ba_is_sentinel(hv) && errw) __hip_atomic_store(errw, (T)BA_DEVERR_SWEEP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        zs[tid] += hv;
    }
    __syncthreads();
    // the group's own unknowns: x1 = W1^T z1;  z0 -= L10^T x1;  x0 = W0^T z0
    T a = 0;
    if (nbg == 2) {
#pragma unroll
        for (int t = 0; t < 16; t++) a += w1[t] * zs[NB + 16 * q + t];
        part[q][j] = a;
        __syncthreads();
        if (tid < NB) xs[NB + tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        __syncthreads();
        a = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) a += l10[t] * xs[NB + 16 * q + t];
        part[q][j] = a;
        __syncthreads();
        if (tid < NB) zs[tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        __syncthreads();
    }
    a = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) a += w0[t] * zs[16 * q + t];
    part[q][j] = a;
    __syncthreads();
    if (tid < NB) xs[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    __syncthreads();
    if (tid < ncg && c0 + tid < ncols) __hip_atomic_store(&x[c0 + tid], xs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Host side of the backward sweep on `st`.  armed: x already holds the sentinel (the solver's k_post_reduce does that).
template <typename T, int NB> inline void ba_ldlt_backsweep_launches(hipStream_t st, int ncols, int ld, int zrow, T *S, const T *Winv, T *x);

// max_groups: beyond that many workgroups (two per group) the sweep falls back to a launch per pair of block columns.  This cap --
// callers pass the CU count, one 256-thread workgroup per CU -- IS the correctness condition of the single launch: with the whole
// grid resident every wait is for a running workgroup.  That k_ldlt_backflow's waits all point at LOWER workgroup indices only helps
// under the ASSUMPTION that workgroups are dispatched in index order, which is what the hardware is observed to do and what HIP does
// not promise (MI355X_MICROARCH.md, "Workgroup dispatch": HIP promises nothing about dispatch order); it is not relied on.  A wait
// that runs out all the same (a GPU shared with another process) raises BA_DEVERR_SWEEP and ba_minimize repeats the trial with one
// launch per pair (ba_solver_try_step returns BA_ERR_HIP: the step-level seam has no retry).
// zh: room for 128 scalars per group (the helpers' partial sums), armed with the sentinel like x.
template <typename T, int NB>
inline void ba_ldlt_backsweep(hipStream_t st, int ncols, int ld, int zrow, T *S, const T *Winv, T *x, T *zh, bool armed = false, int max_groups = 256,
                              T *errw = nullptr, bool safe = false)
{
    const int nblk = (ncols + NB - 1) / NB, groups = (nblk + 1) / 2;
    if (safe || 2 * groups > max_groups) { ba_ldlt_backsweep_launches<T, NB>(st, ncols, ld, zrow, S, Winv, x); return; }
    if (!armed) {
        hipLaunchKernelGGL((k_fill_sentinel<T>), dim3((ncols + 255) / 256), dim3(256), 0, st, ncols, x);
        hipLaunchKernelGGL((k_fill_sentinel<T>), dim3((groups * 2 * NB + 255) / 256), dim3(256), 0, st, groups * 2 * NB, zh);
    }
    hipLaunchKernelGGL((k_ldlt_backflow<T, NB>), dim3(2 * groups), dim3(256), 0, st, ncols, ld, zrow, nblk, S, Winv, x, zh, errw, (int)BA_SWEEP_SPINS);
}

// Host side of the factorisation on `st`: one k_ldlt_panel launch for the first block column, then one fused k_ldlt_step per
// block column (or panel + update launches for a single block column).  flags: nflags ints (hand-off flags of the row
// workgroups), Wp: 2 * ld * NB (double-buffered Y = L D panel), Winv: one NB x NB inverse per block column.
// safe: panel + update launches per block column (no workgroup waits for another one of its launch) -- the retry path after a
// hand-off time-out; fault: self-test, the row workgroups of the fused steps stay silent.
// Side stream of the factorisation (update-bound sizes): a pair's macro tiles run on st2 beside the panel step of the same block column
// (fork / join by events: valid inside a stream capture).  All null: everything on `st`.
struct ba_ldlt_side {
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool on() const { return st2 && ev_fork && ev_join; }
};

template <typename T, int NB>
inline void ba_ldlt_factor(hipStream_t st, int nrows, int ncols, int ld, T *S, T *Wp, T *Winv, int *flags, int nflags, T *errw = nullptr,
                           bool safe = false, int fault = 0, const ba_ldlt_side &side = ba_ldlt_side())
{
    const int nblk = (ncols + NB - 1) / NB;
    const size_t wsz = (size_t)ld * NB;
    const bool k128 = nblk >= 48; // update-bound sizes: two panels per pass over the trailing matrix (Wp: FOUR panels, p % 4)
    const int pair_min = getenv("BA_LDLT_PAIR_MIN") ? atoi(getenv("BA_LDLT_PAIR_MIN")) : 72;
    for (int p = 0; p < nblk; p++) {
        const int p0 = p * NB;
        const int below = nrows - (p0 + NB);
        const int npanel = below > 0 ? (below + 63) / 64 : 1;
        T *wcur = Wp + (size_t)(k128 ? p % 4 : (p & 1)) * wsz, *wprev = Wp + (size_t)(k128 ? (p + 3) % 4 : ((p + 1) & 1)) * wsz;
        // The fused look-ahead step wins at every size (dense bench, D = 100 ... 9216): it saves a launch per block column
        // and keeps the previous panel's update off the diagonal block's path.
        const bool fused = nblk >= 2 && !safe;
        if (p == 0 || !fused) {
            hipLaunchKernelGGL((k_ldlt_panel<T, NB>), dim3(npanel), dim3(256), 0, st, nrows, ncols, ld, p0, S, fused ? wcur : Wp,
                               Winv + (size_t)p * NB * NB, flags, nflags);
            const int p1 = p0 + NB;
            if (!fused && p1 < ncols) {
                const int nti = (nrows - p1 + 63) / 64, ntj = (ncols - p1 + 63) / 64;
                hipLaunchKernelGGL((k_ldlt_update<T, NB>), dim3(ntj, nti), dim3(256), 0, st, nrows, ncols, ld, p0, S, Wp);
            }
        } else {
            // trailing tiles of block column p0 - 64 outside block column p0: rows p0 + 64 ti, cols p0 + 64 tj, 1 <= tj <= ti
            const int nt = (nrows - p0 + 63) / 64, ntc = (ncols - p0 + 63) / 64;
            int nupd = 0;
            for (int ti = 1; ti < nt; ti++) nupd += ti < ntc - 1 ? ti : ntc - 1;
            // Up to D ~ 3000 the panel is the critical path: the variant with the look-ahead update inlined into the
            // sub-panel loop (one workgroup per CU by its dynamic-LDS request, so a panel workgroup never shares its CU
            // with an update workgroup).  Beyond, the update dominates: out-of-line variant, <= 256 registers + < 80 KiB
            // LDS = two per CU (tests/test_kernel_resources.py pins that).
            if (!k128) {
                const int nq = below > 0 ? npanel : 0;      // workgroups that update the panel workgroups' rows (see k_ldlt_step)
                const int np2 = below > 0 ? 2 * npanel : 1; // two panel workgroups per 64-row block (32 rows of the row GEMM each)
                // (the first three steps of a 37-block factorisation are update-bound -- 20 us against the 14 us of the panel chain:
                // there the workgroups may share a CU; 0.545 -> 0.540 ms at D = 2313, and slower again from step 6 on)
                const unsigned dyn_lds = (p <= 3 && nblk >= 32) ? 0 : 8192;
                hipLaunchKernelGGL((k_ldlt_step<T, NB, true>), dim3(nq + np2 + nupd), dim3(256), dyn_lds, st, nrows, ncols, ld, p0, np2, S, wcur,
                                   wprev, Winv + (size_t)p * NB * NB, nq, flags, errw, fault);
            } else {
                // The panels go in pairs (0, 1), (2, 3), ...: an odd step p applies panel p - 1 to the next block column alone (the
                // one step p + 1 factors), the even step p + 1 applies the pair (p - 1, p) to everything from block column p + 2 on
                // with 128 x 128 macro tiles: the trailing matrix is read and written once per two panels.  While fewer than
                // pair_min block columns remain the update no longer sets the pace and a macro tile of a pair (20 us on its own)
                // would outlast the panel (15 us): from the first odd step there, every step applies its predecessor's panel to
                // everything with 64 x 64 tiles.  (Measured, D = 9216: 9.4 ms with the quadrant update and K = 128, 8.0 ms now;
                // pair_min 40 / 56 / 72 / 88 / 104: 8.23 / 8.13 / 8.03 / 8.12 / 8.29 ms.  The same macro tiles with ONE panel in the
                // tail are no faster than the quadrants, 1.80 against 1.75 ms at D = 4608: K = 64 moves 8 flop per byte of C and
                // is bound by that traffic, ~5 TB/s, either way.)
                const auto macro_count = [&](int base) {
                    const int mr = ((nrows - base + 63) / 64 + 1) / 2, mc = ((ncols - base + 63) / 64 + 1) / 2;
                    int n = 0;
                    for (int mi = 0; mi < mr; mi++) n += mi + 1 < mc ? mi + 1 : mc;
                    return base < ncols ? n : 0;
                };
                const int p_single = ((nblk - pair_min) | 1) > 1 ? ((nblk - pair_min) | 1) : 1; // first odd step of the single-panel tail
                int mode = 0, n64 = 0;
                ba_macro_job<T> mj{nullptr, nullptr, 0, 0, 0};
                if (p >= p_single) n64 = nupd;
                else if (p & 1) {
                    mode = 1;
                    n64 = ntc > 1 ? nt - 1 : 0;
                } else mj = {Wp + (size_t)((p - 2) % 4) * wsz, wprev, p0 - 2 * NB, p0 + NB, macro_count(p0 + NB)};
                const int nq = below > 0 ? npanel : 0, np2 = below > 0 ? 2 * npanel : 1;
                static const bool macro8 = getenv("BA_LDLT_MACRO8") != nullptr && atoi(getenv("BA_LDLT_MACRO8")) != 0;
                if (macro8 && mj.count) { // experiment: the pair's macro tiles as a launch of their own, eight waves per tile
                    ba_macro_job<T> none{nullptr, nullptr, 0, 0, 0};
                    // (side stream: the macro tiles beside the panel step -- disjoint parts of S, both behind the previous step; the panel
                    // step is enqueued first so that its workgroups are placed first)
                    if (side.on()) (void)hipEventRecord(side.ev_fork, st);
                    hipLaunchKernelGGL((k_ldlt_step2<T, NB>), dim3(nq + np2 + n64), dim3(256), 0, st, nrows, ncols, ld, p0, np2, S, wcur,
                                       wprev, Winv + (size_t)p * NB * NB, nq, flags, errw, mode, n64, none, fault);
                    if (side.on()) {
                        (void)hipStreamWaitEvent(side.st2, side.ev_fork, 0);
                        hipLaunchKernelGGL((k_ldlt_macro8<T, NB>), dim3(mj.count), dim3(512), 0, side.st2, nrows, ncols, ld, mj, S);
                        (void)hipEventRecord(side.ev_join, side.st2);
                        (void)hipStreamWaitEvent(st, side.ev_join, 0);
                    } else hipLaunchKernelGGL((k_ldlt_macro8<T, NB>), dim3(mj.count), dim3(512), 0, st, nrows, ncols, ld, mj, S);
                    continue;
                }
                hipLaunchKernelGGL((k_ldlt_step2<T, NB>), dim3(nq + np2 + n64 + mj.count), dim3(256), 0, st, nrows, ncols, ld, p0, np2, S, wcur,
                                   wprev, Winv + (size_t)p * NB * NB, nq, flags, errw, mode, n64, mj, fault);
            }
        }
    }
}

// The same with one launch per pair of block columns (k_ldlt_backpair): no workgroup waits for another.
template <typename T, int NB>
inline void ba_ldlt_backsweep_launches(hipStream_t st, int ncols, int ld, int zrow, T *S, const T *Winv, T *x)
{
    const int nblk = (ncols + NB - 1) / NB;
    int p = nblk - 1;
    for (; p >= 1; p -= 2) {
        const int pb = p - 1;
        int g = pb; // 64 earlier columns per workgroup
        if (g < 1) g = 1;
        hipLaunchKernelGGL((k_ldlt_backpair<T, NB>), dim3(g), dim3(256), 0, st, ncols, ld, zrow, pb, S, Winv, x);
    }
    if (p == 0) hipLaunchKernelGGL((k_ldlt_backstep<T, NB>), dim3(1), dim3(256), 0, st, ncols, ld, zrow, 0, S, Winv, x);
}

#endif
