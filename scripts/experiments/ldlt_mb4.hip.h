// scripts/experiments/ldlt_mb4.hip.h -- NOT part of the product: the 4-pivot micro-block factorisation of the 16 x 16 diagonal
// tile that VERDICT r2 item 1 asked for, as it was measured in round 3 (profiles/EXPERIMENTS.md, table "k_ldlt_step: micro-blocks").
// It was spliced into ba_panel_body (wave 0, fp64, full tiles) in place of the pivot-by-pivot loop; results were correct
// (residual 3.9e-15 at D = 2313) and SLOWER: 3.4 - 3.8 k cycles per tile against 2.7 k.  A wave issues one fp64 instruction every
// ~9 cycles; the redundant in-lane 4 x 4 factor + row substitution + selects are ~150 instructions per micro-block (~62 on the
// chain even without the W update), the pivot-by-pivot loop ~25 per pivot: the loop is issue-bound, not exchange-bound.
// Kept for the record; it compiles against ba_dense.hip.h's helpers (ba_rcp, ba_mfma, ba_wave_lds_order, ba_d4).
// ---- the 16 x 16 diagonal tile by 4-pivot micro-blocks (fp64; one wave) ------------------------------------------------
// The tile sits in the registers of ONE wave in the C/D fragment layout of v_mfma_f64_16x16x4: lane (i, q) holds
// T[i][q + 4 v], v = 0..3.  Micro-block m = pivots 4m .. 4m+3: their columns are register m of the four lane groups, i.e. the
// 16 x 4 panel P[i][c] = T[i][4m + c] IS the A / B operand fragment of the instruction (lane (i, c) holds P[i][c]).  One LDS
// trip per micro-block brings the 4 x 4 diagonal block and the lane's own panel row to every lane; the block is factored
// redundantly in every lane -- pivots two at a time in closed form (1/d1 = d0 / (d0 p11 - p10^2): the two reciprocals of a pair
// are independent) -- the multipliers come from an in-lane forward substitution, and the rank-4 update of the whole tile is ONE
// v_mfma_f64_16x16x4 (K = 4 is the instruction's depth).  The tile costs 4 exchanges instead of 15.
// W_ss = L_ss^-1 rides along in the same layout (lane (j, q): W[q + 4 v][j]): eliminating micro-block m multiplies W from the
// left by G_m, which differs from I only in columns 4m .. 4m+3 -- W -= A W[4m .. 4m+3][:] with A[i][c] = sum_k Lm[i][k] M[k][c],
// Lm the multipliers of the micro-block (in-block rows included, zero on and above the diagonal), M the inverse of the 4 x 4
// unit lower factor -- again one MFMA whose B operand is register m of W.
// Writes: Ad[c0 + k][c0 + i] = L[i][k] (i > k), D(k) on the diagonal, 0 above; Wl[c0 + r][c0 + c] = W_ss; dinv[c0 + k] = 1 / D(k).
typedef double ba_d2 __attribute__((ext_vector_type(2)));
template <int NB>
__device__ __forceinline__ void ba_tile_mb4(double (&Ad)[NB][NB + 1], double (&Wl)[NB][NB + 1], double *__restrict__ dinv,
                                            double *__restrict__ stg /* 64 scalars, 16-byte aligned */, int c0, int i, int q, ba_d4 a)
{
    ba_d4 w;
#pragma unroll
    for (int v = 0; v < 4; v++) w[v] = (q + 4 * v == i) ? 1.0 : 0.0;
    // The wave issues in order, so the source order below IS the schedule (sched_barrier between the regions): the W update of
    // micro-block m - 1 fills the wait for the exchange of micro-block m, the stores of m sit behind its MFMA.
    double py0 = 0, py1 = 0, py2 = 0, py3 = 0, pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0, pl10 = 0, pl20 = 0, pl30 = 0, pl21 = 0, pl31 = 0, pl32 = 0;
    auto w_update = [&](int m) { // W <- G_m W from the row (py), reciprocals (pr) and 4 x 4 factor (pl) of micro-block m
        const int r_ = i - 4 * m; // row inside / below the micro-block: multipliers exist for r_ > k (zero above: those rows of W are final)
        const double L0 = r_ > 0 ? py0 * pr0 : 0.0, L1 = r_ > 1 ? py1 * pr1 : 0.0, L2 = r_ > 2 ? py2 * pr2 : 0.0, L3 = r_ > 3 ? py3 * pr3 : 0.0;
        const double M10 = -pl10, M21 = -pl21, M32 = -pl32;
        const double M20 = fma(-pl21, M10, -pl20), M31 = fma(-pl32, M21, -pl31);
        const double M30 = fma(-pl32, M20, fma(-pl31, M10, -pl30));
        const double A0 = fma(L3, M30, fma(L2, M20, fma(L1, M10, L0)));
        const double A1 = fma(L3, M31, fma(L2, M21, L1));
        const double A2 = fma(L3, M32, L2);
        const double asel = q == 0 ? A0 : q == 1 ? A1 : q == 2 ? A2 : L3;
        const double wm = w[m];
        w = ba_mfma(-asel, wm, w);
    };
#pragma unroll
    for (int m = 0; m < 4; m++) {
        // ---- exchange: the panel of the micro-block through LDS (in order inside a wave: no wait between store and loads)
        stg[4 * i + q] = a[m];
        ba_wave_lds_order();
        const double *pd = stg + 16 * m;
        const double p00 = pd[0];
        const ba_d2 p1 = *(const ba_d2 *)(pd + 4), p2 = *(const ba_d2 *)(pd + 8);
        const double p22 = pd[10];
        const ba_d2 p3a = *(const ba_d2 *)(pd + 12), p3b = *(const ba_d2 *)(pd + 14);
        const ba_d2 ra = *(const ba_d2 *)(stg + 4 * i), rb = *(const ba_d2 *)(stg + 4 * i + 2);
        ba_wave_lds_order();
        __builtin_amdgcn_sched_barrier(0);
#ifndef BA_KO_W
        if (m > 0) w_update(m - 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        const double p10 = p1.x, p11 = p1.y, p20 = p2.x, p21 = p2.y, p30 = p3a.x, p31 = p3a.y, p32 = p3b.x, p33 = p3b.y;
        // ---- 4 x 4 LDL^T, pivots (0, 1) and (2, 3) in closed form
        const double d0 = p00;
        const double det01 = fma(d0, p11, -(p10 * p10));
        const double r0 = ba_rcp(d0), rdet01 = ba_rcp(det01);
        const double r1 = d0 * rdet01;
        const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
        const double y21 = fma(-l20, p10, p21), y31 = fma(-l30, p10, p31);
        const double l21 = y21 * r1, l31 = y31 * r1;
        const double d2 = fma(-l21, y21, fma(-l20, p20, p22));
        const double y32 = fma(-l31, y21, fma(-l30, p20, p32));
        const double q33 = fma(-l31, y31, fma(-l30, p30, p33));
        const double det23 = fma(d2, q33, -(y32 * y32));
        const double r2 = ba_rcp(d2), rdet23 = ba_rcp(det23);
        const double r3 = d2 * rdet23;
        const double l32 = y32 * r2;
        // ---- this lane's row of the panel: Y = P L_d^-T, L = Y D^-1; lane (i, q) feeds column q to the matrix cores
        const double y0 = ra.x;
        const double y1 = fma(-l10, y0, ra.y);
        const double y2 = fma(-l21, y1, fma(-l20, y0, rb.x));
        const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, rb.y)));
        const double ys012 = q == 0 ? y0 : q == 1 ? y1 : y2, rs012 = q == 0 ? r0 : q == 1 ? r1 : r2;
        const double ysel = q == 3 ? y3 : ys012, rsel = q == 3 ? r3 : rs012;
        const double lsel = ysel * rsel;
        // T[j][i'] -= L[j][k] Y[i'][k]: rows j on and above the micro-block are registers 0 .. m of the tile, dead from here on, so
        // the operand needs no mask
        if (m < 3) a = ba_mfma(-lsel, ysel, a);
        __builtin_amdgcn_sched_barrier(0);
        // ---- results: L(., 4m + q) (whatever lands on and above the diagonal of the tile is never read); D and 1 / D by lane 0
        Ad[c0 + 4 * m + q][c0 + i] = lsel;
        if (i + 16 * q == 0) {
            const double d1 = fma(-l10, p10, p11), d3 = fma(-l32, y32, q33);
            Ad[c0 + 4 * m + 0][c0 + 4 * m + 0] = d0;
            Ad[c0 + 4 * m + 1][c0 + 4 * m + 1] = d1;
            Ad[c0 + 4 * m + 2][c0 + 4 * m + 2] = d2;
            Ad[c0 + 4 * m + 3][c0 + 4 * m + 3] = d3;
            dinv[c0 + 4 * m + 0] = r0;
            dinv[c0 + 4 * m + 1] = r1;
            dinv[c0 + 4 * m + 2] = r2;
            dinv[c0 + 4 * m + 3] = r3;
        }
        ba_wave_lds_order();
        py0 = y0; py1 = y1; py2 = y2; py3 = y3; pr0 = r0; pr1 = r1; pr2 = r2; pr3 = r3;
        pl10 = l10; pl20 = l20; pl30 = l30; pl21 = l21; pl31 = l31; pl32 = l32;
    }
#ifndef BA_KO_W
    w_update(3);
#endif
#pragma unroll
    for (int v = 0; v < 4; v++) Wl[c0 + q + 4 * v][c0 + i] = (i <= q + 4 * v) ? w[v] : 0.0;
}

