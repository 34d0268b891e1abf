// scripts/experiments/ldlt_small.hip.h -- EXPERIMENT, not part of the product (profiles/EXPERIMENTS.md 1.5: measured, slower than the
// 64-wide path it was meant to replace).  K6 for SMALL reduced camera systems (D <= 207): the LDL^T of ba_dense.hip.h, the forward elimination of the
// right-hand side and the backward sweep in ONE launch of ONE workgroup.
//
// Stands in for the same reference calls as ba_dense.hip.h (Eigen::SimplicialLDLT on the reduced camera matrix,
// src/Eigen_ext/BacktrackLevMarqQRChol.h:339-341, src/Eigen_ext/BacktrackLevMarqCholesky.h:274-282) when the matrix is a few tiles
// wide (configs 1 and 2: D = 144, 189).  There the 64-wide machinery costs three launches of ~14 us + the sweep's launch for a matrix
// whose whole lower triangle fits the register file of one CU.
//
// Layout: the lower triangle in 16 x 16 tiles; tile t (column-major over tile columns) lives in REGISTERS of wave 1 + t % 7, slot
// t / 7, in the C/D fragment layout of the 16x16x4 matrix-core instruction, transposed: register v of lane (li, lk) holds
// A[16 ti + li][16 tj + crow(lk, v)].  Wave 0 owns no tile: it runs the critical chain, and never waits at a workgroup barrier
// while the factorisation runs -- the eight waves hand their results over through flags in LDS (bounded waits):
//   wave 0, step s:  pivot loop on the diagonal tile (dg[s & 1], the loop of ba_panel_body) -> W_ss = L_ss^-1, 1 / D   [piv_done]
//                    Y^T = W_ss X^T for the tile below it (pan[s & 1]) and with it step s applied to the NEXT diagonal tile
//                    (dg[(s + 1) & 1]): everything pivot loop s + 1 needs                                              [ysub_done]
//   owners, step s:  (after piv_done) Y for their tiles of tile column s (Y replaces X in the panel; L = Y D^-1 stays in the tile's
//                    registers, which nothing else needs any more: the backward sweep reads L from there); a barrier among the seven
//                    owners; then step s applied to their tiles in the order the chain needs them -- tile (s + 2, s + 1) first
//                    (copied out as the head of the next panel [sub_ready]), then the diagonal tile (s + 2, s + 2) (copied out with
//                    steps <= s applied [diag_ready]; wave 0 adds step s + 1), then the rest of tile column s + 1 (the next panel)
//                    and everything to the right of it.
// The right-hand side rides along as matrix row D (ba_dense.hip.h); after the last step the sweep x = L^-T z runs over the tile
// rows from the bottom: every wave forms x_b = W_bb^T z_b itself (LDS operands), the owners of tile row b subtract L_ba^T x_b from
// z_a (a row reduction by DPP inside the tile).  Nothing but x is written to memory.
#ifndef BA_SMALL_HIP_H
#define BA_SMALL_HIP_H

#include "ba_dense.hip.h"

#define BA_SM_NT 13                       /* tile rows at most: D + 1 <= 208 (91 tiles: 13 slots of 4 registers per owner wave) */
#define BA_SM_OWNERS 7                    /* waves 1..7 */
#define BA_SM_SLOTS 13                    /* 91 / 7 tiles per owner wave */
#define BA_SM_RP (16 * BA_SM_NT + 1)      /* pitch of a panel column */
#define BA_SM_MAXD (16 * BA_SM_NT - 1)

// sum over the 16 lanes of a DPP row (every lane of the row gets it)
template <typename T> __device__ __forceinline__ T ba_row16_sum(T v)
{
    v = ba_dpp_add<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
    v = ba_dpp_add<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
    v = ba_dpp_add<0x141, 0xf>(v); // row_half_mirror
    v = ba_dpp_add<0x140, 0xf>(v); // row_mirror
    return v;
}

// Diagnostic build only (-DBA_SMALL_STAMP, scripts/bench_small.hip): cycle stamps of wave 0 (pivot loop, tail) and wave 1 per step.
#ifdef BA_SMALL_STAMP
__device__ long long ba_small_stamp[2][16][6];
#define BA_SM_STAMP(j) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (wv < 2 && lane == 0) ba_small_stamp[wv][s][j] = (long long)t_; }
#define BA_SM_STAMP_AT(sidx, j) { const int s = sidx; BA_SM_STAMP(j) }
#else
#define BA_SM_STAMP(j)
#define BA_SM_STAMP_AT(sidx, j)
#endif

// body(integral_constant<int, u>) for the slots ua <= u < ub: ONE computed jump into the unrolled sequence (a tile's registers can
// only be named by a compile-time slot index; testing every slot for every phase instead cost ~60 cycles of instruction fetch per
// skipped slot: 2 400 cycles per phase for a wave with nothing to do)
template <int K> struct ba_sm_slot { static constexpr int value = K; };
template <typename F> __device__ __forceinline__ void ba_sm_for_slots(int ua, int ub, F &&body)
{
    static_assert(BA_SM_SLOTS == 13, "one case per slot");
#define BA_SM_CASE(k) case k: if (k >= ub) break; body(ba_sm_slot<k>()); [[fallthrough]];
    switch (ua) {
        BA_SM_CASE(0) BA_SM_CASE(1) BA_SM_CASE(2) BA_SM_CASE(3) BA_SM_CASE(4) BA_SM_CASE(5) BA_SM_CASE(6)
        BA_SM_CASE(7) BA_SM_CASE(8) BA_SM_CASE(9) BA_SM_CASE(10) BA_SM_CASE(11) BA_SM_CASE(12)
    default: break;
    }
#undef BA_SM_CASE
}

#define BA_SM_SPINS (1 << 20) /* bound of a hand-off wait (x ~100 cycles); a wait that runs out is reported through errw */
enum { BA_SM_PIV = 0, BA_SM_YSUB, BA_SM_DIAG, BA_SM_SUB, BA_SM_OCNT, BA_SM_ERR, BA_SM_NSYNC };

template <typename T>
__global__ __launch_bounds__(512) void k_ldlt_small(int D, int ld, const T *__restrict__ S, T *__restrict__ x, T *__restrict__ errw = nullptr)
{
    typedef typename ba_acc<T>::type acc_t;
    __shared__ T pan[2][16][BA_SM_RP];     // panel of step s: pan[s & 1][column][row - 16 s], rows 16.. (X, then Y = L D)
    __shared__ T dg[2][16][17];            // diagonal tile of step s: dg[s & 1][column][row]
    __shared__ T Wt[BA_SM_NT][16][17];     // W_ss = L_ss^-1 per step, [row][column]
    __shared__ T dinv[16 * BA_SM_NT], zbuf[16 * BA_SM_NT];
    __shared__ T colx4[4][16], junkbuf[64];
    __shared__ int sy[BA_SM_NSYNC];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const int nct = (D + 15) >> 4;   // tile columns (steps)
    const int ntr = (D >> 4) + 1;    // tile rows: the rhs row D is the last row
    BA_SM_STAMP_AT(15, 0)
    if (tid < BA_SM_NSYNC) sy[tid] = 0;
    // ---- this wave's tiles
    int sti[BA_SM_SLOTS], stj[BA_SM_SLOTS];
    acc_t acc[BA_SM_SLOTS];
#pragma unroll
    for (int u = 0; u < BA_SM_SLOTS; u++) {
        int rem = BA_SM_OWNERS * u + (wv - 1), tj = 0;
        if (wv == 0) rem = 1 << 20;
        while (tj < nct && rem >= ntr - tj) { rem -= ntr - tj; tj++; }
        const bool ok = tj < nct;
        sti[u] = ok ? tj + rem : -1; // (no tile: ti = tj = -1 matches no step)
        stj[u] = ok ? tj : -1;
    }
#pragma unroll
    for (int u = 0; u < BA_SM_SLOTS; u++) {
        if (stj[u] < 0) continue; // (uniform)
        const int r = 16 * sti[u] + li;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int c = 16 * stj[u] + ba_crow<T>(lk, v);
            // (diagonal tiles: the lower triangle mirrored into a full tile; 32-bit offsets from the uniform base: the 52 loads of a
            // lane are in flight together, and a 64-bit address each would cost as many registers as the tiles themselves)
            acc[u][v] = S[(unsigned)(min(r, c) * ld + max(r, c))];
        }
    }
    for (int i = tid; i < 16 * BA_SM_NT; i += 512) { zbuf[i] = (T)0; dinv[i] = (T)0; }
    // panel 0 and the first two diagonal tiles
#pragma unroll
    for (int u = 0; u < BA_SM_SLOTS; u++) {
        if (stj[u] == 0 && sti[u] > 0) {
#pragma unroll
            for (int v = 0; v < 4; v++) pan[0][ba_crow<T>(lk, v)][16 * sti[u] + li] = acc[u][v];
        }
        if (stj[u] == sti[u] && (stj[u] == 0 || stj[u] == 1)) {
#pragma unroll
            for (int v = 0; v < 4; v++) dg[stj[u]][ba_crow<T>(lk, v)][li] = acc[u][v];
        }
    }
    __syncthreads();
    BA_SM_STAMP_AT(15, 1)
    if (tid == 0) { sy[BA_SM_DIAG] = 2; sy[BA_SM_SUB] = 1; }
    __syncthreads();
    // ---- hand-offs (LDS, workgroup scope)
    auto wait_ge = [&](int idx, int want) {
        int spins = 0;
        while (__hip_atomic_load(&sy[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
            if (++spins > BA_SM_SPINS || __hip_atomic_load(&sy[BA_SM_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                __hip_atomic_store(&sy[BA_SM_ERR], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // every later wait falls through: the launch ends
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto post = [&](int idx, int val) { // (release: the wave's LDS stores are complete before the flag moves)
        if (lane == 0) __hip_atomic_store(&sy[idx], val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // step `sp` applied to this wave's slot u (operands: the panel of step sp; dk: 1 / D of the step's pivots 4 kk + lk)
    auto apply = [&](int u, int sp, const T (&dk)[4]) {
        const T(*P)[BA_SM_RP] = pan[sp & 1];
        T la[4], yb[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            la[kk] = P[4 * kk + lk][16 * (stj[u] - sp) + li]; // A[j][k] = L[j][k] = Y[j][k] / D(k)
            yb[kk] = P[4 * kk + lk][16 * (sti[u] - sp) + li]; // B[k][i] = Y[i][k]
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc[u] = ba_mfma(-(la[kk] * dk[kk]), yb[kk], acc[u]);
    };
    // tiles are numbered column by column; this wave's slot u holds tile 7 u + wv - 1: a range of tile numbers is a range of slots
    const int wo = wv - 1;
    auto toff = [&](int c) { return c * ntr - ((c * (c - 1)) >> 1); }; // tiles in the tile columns before c
    auto slot_lo = [&](int t) { const int d = t - wo; return d <= 0 ? 0 : (d + BA_SM_OWNERS - 1) / BA_SM_OWNERS; }; // first slot with tile number >= t
    auto for_tiles = [&](int t0, int t1, auto &&body) { ba_sm_for_slots(slot_lo(t0), min(slot_lo(t1), BA_SM_SLOTS), body); };
    if (wv == 0) {
        // =============================== the critical chain
#pragma unroll 1
        for (int s = 0; s < nct; s++) {
            T(*P)[BA_SM_RP] = pan[s & 1];
            T(*G)[17] = dg[s & 1];
            const int np = min(16, D - 16 * s); // pivots of this step
            BA_SM_STAMP(0)
            wait_ge(BA_SM_DIAG, s + 1);
            BA_SM_STAMP(1)
            const int i = li, q = lk;
            T a[4];
#pragma unroll
            for (int c = 0; c < 4; c++) a[c] = G[4 * q + c][i];
            ba_wave_lds_order();
            T *const junk = junkbuf + lane;
            T lprev = (T)0;
            T w[4];
#pragma unroll
            for (int c = 0; c < 4; c++) w[c] = (4 * q + c == i) ? (T)1 : (T)0;
            auto lstore = [&](int k) {
                *((q == (k >> 2)) ? &G[k][i] : junk) = lprev;
                T wk[4];
#pragma unroll
                for (int c = 0; c < 4; c++) wk[c] = ba_rowbcast_k(w[c], k);
#pragma unroll
                for (int c = 0; c < 4; c++) w[c] -= lprev * wk[c];
            };
            auto pivot = [&](int k) {
                const int kq = k >> 2, kc = k & 3;
                colx4[q][i] = a[kc];
                ba_wave_lds_order();
                const T lraw = colx4[kq][i];
                T y[4];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                ba_wave_lds_order();
                if (k > 0) lstore(k - 1);
                ba_wave_lds_order();
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                const T r = ba_rcp(dk);
                __builtin_amdgcn_sched_barrier(0);
                int iv = i;
                asm volatile("" : "+v"(iv));
                const T lm = (iv > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                lprev = l;
            };
            if (np == 16) {
#pragma unroll
                for (int k = 0; k < 15; k++) pivot(k);
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
                Wt[s][i][j] = (j <= i) ? w[c] : (T)0;
            }
            {
                const int ic = i & 3;
                const T dsel = (ic == 0) ? a[0] : (ic == 1) ? a[1] : (ic == 2) ? a[2] : a[3];
                if (q == (i >> 2) && i < np) dinv[16 * s + i] = ba_rcp(dsel);
            }
            post(BA_SM_PIV, s + 1);
            BA_SM_STAMP(2)
            if (s + 1 < ntr) { // the tile below: Y^T = W_ss X^T, and step s on the next diagonal tile
                wait_ge(BA_SM_SUB, s + 1);
                BA_SM_STAMP(3)
                T wa[4], xb[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    wa[kk] = Wt[s][li][4 * kk + lk]; // A[j][k] = W_ss[j][k]
                    xb[kk] = P[4 * kk + lk][16 + li]; // B[k][n] = X[n][k]
                }
                acc_t y;
#pragma unroll
                for (int v = 0; v < 4; v++) y[v] = 0;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) y = ba_mfma(wa[kk], xb[kk], y);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int j = ba_crow<T>(lk, v);
                    P[j][16 + li] = j < np ? y[v] : (T)0; // columns past the last pivot do not take part
                }
                ba_wave_lds_order();
                if (s + 1 < nct) {
                    wait_ge(BA_SM_DIAG, s + 2);
                    T(*Gn)[17] = dg[(s + 1) & 1];
                    T la[4], yb[4];
                    acc_t c2;
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        yb[kk] = P[4 * kk + lk][16 + li];
                        la[kk] = yb[kk] * dinv[16 * s + 4 * kk + lk];
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) c2[v] = Gn[ba_crow<T>(lk, v)][li];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) c2 = ba_mfma(-la[kk], yb[kk], c2);
#pragma unroll
                    for (int v = 0; v < 4; v++) Gn[ba_crow<T>(lk, v)][li] = c2[v];
                    ba_wave_lds_order();
                }
                post(BA_SM_YSUB, s + 1);
            }
            BA_SM_STAMP(4)
        }
    } else {
        // =============================== the tile owners
#pragma unroll 1
        for (int s = 0; s < nct; s++) {
            T(*P)[BA_SM_RP] = pan[s & 1];
            T(*Pn)[BA_SM_RP] = pan[(s + 1) & 1];
            const int np = min(16, D - 16 * s);
            BA_SM_STAMP(0)
            wait_ge(BA_SM_PIV, s + 1);
            BA_SM_STAMP(1)
            T dk[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) dk[kk] = dinv[16 * s + 4 * kk + lk];
            // ---- Y for this wave's tiles of tile column s (the one right below the diagonal tile is wave 0's)
            const int c0 = toff(s), c1 = toff(s + 1), c2 = toff(s + 2), nt = toff(nct);
            for_tiles(c0 + 2, c1, [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int r0 = 16 * (sti[u] - s);
                T wa[4], xb[4], dv[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    wa[kk] = Wt[s][li][4 * kk + lk];
                    xb[kk] = P[4 * kk + lk][r0 + li];
                }
                acc_t y;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    y[v] = 0;
                    dv[v] = dinv[16 * s + ba_crow<T>(lk, v)];
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) y = ba_mfma(wa[kk], xb[kk], y);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int j = ba_crow<T>(lk, v);
                    const T yv = j < np ? y[v] : (T)0;
                    P[j][r0 + li] = yv;
                    acc[u][v] = yv * dv[v];
                }
            });
            // ---- every owner's Y (and wave 0's) is an operand of the updates below
            if (lane == 0) __hip_atomic_fetch_add(&sy[BA_SM_OCNT], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            wait_ge(BA_SM_OCNT, BA_SM_OWNERS * (s + 1));
            if (s + 1 < ntr) wait_ge(BA_SM_YSUB, s + 1);
            BA_SM_STAMP(2)
            if (s + 1 < ntr)
                for_tiles(c0 + 1, c0 + 2, [&](auto uc) { // wave 0 formed this tile's Y: L = Y D^-1 for the sweep
                    constexpr int u = decltype(uc)::value;
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[u][v] = P[ba_crow<T>(lk, v)][16 + li] * dinv[16 * s + ba_crow<T>(lk, v)];
                });
            if (s + 1 < nct) {
                // ---- step s on this wave's tiles, the ones wave 0 waits for first: (s + 2, s + 1), then the diagonal tile (s + 2, s + 2)
                if (s + 2 < ntr)
                    for_tiles(c1 + 1, c1 + 2, [&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        apply(u, s, dk);
#pragma unroll
                        for (int v = 0; v < 4; v++) Pn[ba_crow<T>(lk, v)][16 + li] = acc[u][v];
                        post(BA_SM_SUB, s + 2);
                    });
                if (s + 2 < nct)
                    for_tiles(c2, c2 + 1, [&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        apply(u, s, dk);
#pragma unroll
                        for (int v = 0; v < 4; v++) dg[s & 1][ba_crow<T>(lk, v)][li] = acc[u][v];
                        post(BA_SM_DIAG, s + 3);
                    });
                BA_SM_STAMP(3)
                for_tiles(c1 + 2, c2, [&](auto uc) { // the rest of tile column s + 1: the next panel
                    constexpr int u = decltype(uc)::value;
                    apply(u, s, dk);
#pragma unroll
                    for (int v = 0; v < 4; v++) Pn[ba_crow<T>(lk, v)][16 * (sti[u] - s - 1) + li] = acc[u][v];
                });
                for_tiles(c2 + 1, nt, [&](auto uc) { apply(decltype(uc)::value, s, dk); }); // everything to the right of it
            }
            BA_SM_STAMP(4)
        }
    }
    BA_SM_STAMP_AT(15, 2)
    __syncthreads();
    BA_SM_STAMP_AT(15, 3)
    if (errw && tid == 0 && sy[BA_SM_ERR]) __hip_atomic_store(errw, (T)BA_DEVERR_SWEEP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- z = D^-1 L^-1 b: row D of L
    const int tr = D >> 4, zr = D & 15;
    if (zr != 0) { // the rhs row runs through the last diagonal tile: its multipliers are there
        if (tid < zr) zbuf[16 * tr + tid] = dg[(nct - 1) & 1][tid][zr];
    }
    if (wv > 0)
        for (int a = 0; a < tr && a < nct; a++) { // tiles (tr, a): row zr of each is z
            const int t = toff(a) + tr - a;
            if ((t - wo) % BA_SM_OWNERS == 0)
                for_tiles(t, t + 1, [&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    if (li == zr) {
#pragma unroll
                        for (int v = 0; v < 4; v++) zbuf[16 * a + ba_crow<T>(lk, v)] = acc[u][v];
                    }
                });
        }
    // ---- x = L^-T z, tile row by tile row from the bottom
#pragma unroll 1
    for (int b = nct - 1; b >= 0; b--) {
        __syncthreads();
        // x_b = W_bb^T z_b: lane (li, lk) sums rows 4 lk .. 4 lk + 3 of column li, then the four parts
        T xp = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) xp += Wt[b][4 * lk + c][li] * zbuf[16 * b + 4 * lk + c];
        xp += __shfl_xor(xp, 16, 64);
        xp += __shfl_xor(xp, 32, 64);
        const int np = min(16, D - 16 * b);
        const T xb = li < np ? xp : (T)0;
        if (wv == 1 && lk == 0 && li < np) x[16 * b + li] = xb;
        if (wv > 0)
            for (int a = 0; a < b; a++) { // z_a -= L_ba^T x_b by the owner of tile (b, a)
                const int t = toff(a) + b - a;
                if ((t - wo) % BA_SM_OWNERS == 0)
                    for_tiles(t, t + 1, [&](auto uc) {
                        constexpr int u = decltype(uc)::value;
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const T sum = ba_row16_sum(acc[u][v] * xb);
                            if (li == 0) zbuf[16 * a + ba_crow<T>(lk, v)] -= sum;
                        }
                    });
            }
    }
    BA_SM_STAMP_AT(15, 4)
}

#endif
