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
// reduction: level 1 cuts the rows into chunks of CH (1024 fp32 / 512 fp64) that ONE WORKGROUP factors in registers (Householder,
// column by column, reflectors stored in place below the diagonal of the chunk, the chunk's 32 x 32 R on its top rows); level
// L + 1 stacks the R's of NSB = CH / 32 level-L chunks (their top rows, in place: row stride CH * NSB^(L-1)) and factors the
// stack the same way -- its reflectors only have entries where the stacked triangles had them, so they fit in the triangles
// they annihilate and the lower-level reflectors underneath stay intact.  Three levels at config 3 (181 633 rows).
// Every chunk also leaves the T factor of its 32 reflectors (compact WY: H_0 ... H_31 = I - V T V^T), so the trailing columns
// (and the right-hand side, which rides along as column D) receive a chunk's reflectors as THREE PRODUCTS ON THE MATRIX CORES
// -- W = V^T B, Z = T^T W, B -= V Z (k_qr_apply) -- instead of 32 rank-1 updates on the vector units (round 2: 5.8 % of the
// fp32 rate, 110 launches per trial over five levels).
#ifndef BA_QR_HIP_H
#define BA_QR_HIP_H

#include <hip/hip_runtime.h>
#include "ba_mfma.hip.h"

#define BA_QR_PB 32 /* panel width = rows of a sub-block */

// Chunk heights: level 1 -- where the rows are -- takes chunks of NSB = 32 (fp32) / 16 (fp64) sub-blocks of 32 rows (one round of
// workgroups at config 3: 178 chunks); the upper levels stack NSBU = 16 R's per chunk: a few short tasks whose latency is the
// panel's critical path (178 -> 12 -> 1 chunks at config 3).
template <typename T> struct ba_qr_cfg {
    static constexpr int NSB = sizeof(T) == 4 ? 32 : 16; // level 1: sub-blocks per chunk
    static constexpr int CH = BA_QR_PB * NSB;            // rows per level-1 chunk = 64 lanes x 16 (fp32) / 8 (fp64) rows: a wave's CH x 8 columns are 128 registers per lane
    static constexpr int NSBU = 16;                      // upper levels (8: one more level at config 3, 137 us of chain per panel against 90)
};

// global row of local row l of chunk g: sub-block s = l / 32 starts at row0 + (g NSB + s) stride, stride = 32 at level 1
template <int NSB> __device__ __forceinline__ size_t ba_qr_row(int row0, int g, int l, long long stride)
{
    return (size_t)row0 + (size_t)((long long)g * NSB + (l >> 5)) * (size_t)stride + (size_t)(l & 31);
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
                                                     T *__restrict__ A, size_t lda, int cam_rows /* sharded: the camera rows belong to shard 0 */,
                                                     const int *__restrict__ go = nullptr)
{
    if (go && *go == 0) return; // (uniform)
    const int ia = blockIdx.x * 256 + threadIdx.x;
    if (cam_rows && ia < D) A[(size_t)ia * lda + 2 * (size_t)K + 3 * (size_t)Ml + ia] = sqrt(*lam); // camera rows: sqrt(lambda) I_D, zero rhs
    if (ia >= K) return;
    const int j = obs_pt[ia], a = obs_cam[ia], b = pt_ptr[j], e = pt_ptr[j + 1];
    // A camera may see a point more than once (legal input: the reference's sparse J just has more rows): the blocks of those
    // observations ADD UP in the 9 columns of the camera.  The first of them writes the sum (Z summed in observation order), the
    // others write nothing -- no atomics, and the same bits as before where every (point, camera) pair occurs once.
    for (int i2 = b; i2 < ia; i2++)
        if (obs_cam[i2] == a) return;
    const size_t r0 = 2 * (size_t)b + 3 * (size_t)j;
    T Z[27];
#pragma unroll
    for (int q = 0; q < 27; q++) Z[q] = rec[(size_t)ia * BA_REC + q];
    for (int i2 = ia + 1; i2 < e; i2++)
        if (obs_cam[i2] == a) {
#pragma unroll
            for (int q = 0; q < 27; q++) Z[q] += rec[(size_t)i2 * BA_REC + q];
        }
    T *colbase = A + (size_t)(9 * a) * lda;
    for (int ib = b; ib < e; ib++) {
        const T *Q = q1obs + 6 * (size_t)ib; // 2 x 3 row-major
        T q[6];
#pragma unroll
        for (int m = 0; m < 6; m++) q[m] = Q[m];
        const bool mine = obs_cam[ib] == a; // (ib == ia, or another observation of the same camera)
#pragma unroll
        for (int c = 0; c < 9; c++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                T v = mine ? Jc[(size_t)(9 * rr + c) * K + ib] : (T)0;
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

// Entry l of reflector c of chunk g (local row l of the chunk): 1 on the pivot row, zero above it; below it the stored panel column
// at level 1, and at the upper levels -- where every sub-block of 32 rows is the R of a lower-level chunk -- only the rows t <= c of
// the sub-blocks behind the first (the triangle the reflector annihilated; what lies under it is the lower level's reflectors).
// The load is unconditional from a clamped (valid) address and masked afterwards: a conditional load is compiled as branch + wait.
// Two steps, so that a batch of loads can be ISSUED before the first of them is looked at: ba_qr_vraw is the plain load,
// ba_qr_vmask the mask -- with an opaque use of the loaded value in front of it, or the compiler sinks every load under its mask
// again (a branch and a full wait per load: 44 us per k_qr_apply task instead of 10).
template <typename T>
__device__ __forceinline__ T ba_qr_vraw(const T *__restrict__ A, size_t lda, int c0, int bw, size_t grow_l, int c)
{
    return A[(size_t)(c0 + (c < bw ? c : 0)) * lda + grow_l];
}
template <typename T> __device__ __forceinline__ T ba_qr_vmask(T x, int bw, int level, int rows, int l, int c)
{
    asm volatile("" : "+v"(x));
    const bool stored = c < bw && l > c && l < rows && (level == 1 || ((l >> 5) > 0 && (l & 31) <= c));
    return stored ? x : ((l == c && c < bw) ? (T)1 : (T)0);
}

#define BA_QR_TP 36 /* pitch of the per-wave LDS tile [16 rows][32 columns]: fragment reads (lane <-> column, rows 4 q + v) and 16-byte row
                       writes are both free of bank conflicts (16 q mod 32 separates the two halves of a 32-lane access) */
// 16 rows x 32 columns between memory and a wave: lane (i, q) owns row i of the EIGHT columns 8 q .. 8 q + 7 -- per instruction
// (one column per lane) 16 rows of 4 columns = four 64-byte runs; in the LDS tile the lane's eight values are two 16-byte words.
template <typename T> __device__ __forceinline__ void ba_qr_tile_to_lds(T *st, const T (&x)[8], int i, int q)
{
    T *d = st + i * BA_QR_TP + 8 * q;
#pragma unroll
    for (int t = 0; t < 8; t++) d[t] = x[t]; // (contiguous, 16-byte aligned: the compiler merges them)
}
template <typename T> __device__ __forceinline__ void ba_qr_tile_from_lds(const T *st, T (&x)[8], int i, int q)
{
    const T *d = st + i * BA_QR_TP + 8 * q;
#pragma unroll
    for (int t = 0; t < 8; t++) x[t] = d[t];
}

// ---- one chunk of a TSQR level: Householder QR of its rows of the panel, in registers --------------------------------------
// Lane l holds rows l, l + 64, ... of the chunk (RPL = CH / 64 of them); the 32 panel columns are dealt to the eight wavefronts of
// the workgroup cyclically: wave w owns the columns 8 jq + w, jq = 0 .. 3, at the compile-time register positions jq (the step
// loop is unrolled over jq: no register is ever moved).  Step j = 8 jq + jw: wave jw forms the reflector of its column jq (norm
// below the pivot by a wave reduction, the scalars redundantly in every lane), hands it to the others through LDS (double-buffered:
// one barrier per step) and retires the column to memory (R entries above the pivot, beta on it, v below); then every wave updates
// its columns behind j, one wave reduction per column for v . a_c.
// level 1: the chunk's rows are dense; level > 1: every sub-block of 32 rows is an upper triangle (the R of a lower-level chunk) --
// entries below a sub-block's diagonal are read as zero and never written (the lower level's reflectors live there).
// Tail: the T factor of the chunk's reflectors (compact WY, forward columnwise like LAPACK's larft):
//   G = V^T V on the matrix cores (each wave its quarter of the rows, summed through LDS), then
//   T(j, j) = tau_j,  T(0:j, j) = -tau_j T(0:j, 0:j) G(0:j, j)  -- one lane per row of T, 32 dependent steps --
// written row-major to Tout[chunk][32][32] for k_qr_apply.
// Scalars of a reflector.  float: the hardware square root and reciprocal (1 ulp) with one Newton step on the reciprocal -- the IEEE
// sequences the compiler emits for `sqrtf` and `/` are ~40 instructions on the critical path of EVERY reflector step; a reflector's
// beta and tau only have to be consistent with each other to working precision.  double: IEEE (the parity tests run in fp64).
// (BA_QR_HW_SQRT=1, diagnostic: the bare v_sqrt_f32 for beta -- round 3 saw config 3 accept no step with it; round 4's look at
// that is in profiles/EXPERIMENTS.md 6.3)
__device__ int ba_qr_hw_sqrt_flag = 0;
__device__ int ba_qr_dbg_flag = 0; // diagnostic bits (BA_QR_DBG): 1 = full barrier in the step loop, 2 = agent acquire before the T factor re-reads V, 4 = one wave reduction per column, 8 = the column retired in front of the step's barrier
// (x is a normal float here: smaller squared norms were rescaled by the caller.)  hw = 0: v_sqrt_f32 (1 ulp) + one Newton step;
// 1: the bare instruction; 2: sqrtf (IEEE: ~40 dependent instructions, 280 cycles of every reflector step by the in-kernel stamps).
__device__ __forceinline__ float ba_qr_sqrt(float x, int hw)
{
    if (hw == 2) return sqrtf(x);
    const float y = __builtin_amdgcn_sqrtf(x);
    if (hw == 1 || !(y > 0.0f) || !(y < __builtin_inff())) return y; // (0: the instruction flushed a denormal argument; inf: overflowed norm)
    return fmaf(fmaf(-y, y, x), 0.5f * __builtin_amdgcn_rcpf(y), y);
}
__device__ __forceinline__ double ba_qr_sqrt(double x, int) { return sqrt(x); }
// below this alpha^2 + |x|^2 a reflector's norm is formed again from entries scaled by `up` (a power of two: exact)
template <typename T> struct ba_qr_tiny;
// ... and above `big` (or overflowed: fp32 outlier observations reach 1e19 and more) from entries scaled by `dn`
template <> struct ba_qr_tiny<float> { static constexpr float s2 = 0x1p-80f, up = 0x1p100f, down = 0x1p-100f, big = 0x1p100f, dn = 0x1p-70f, undn = 0x1p70f; }; // (entries below 2^-40: even denormal ones become normal)
template <> struct ba_qr_tiny<double> { static constexpr double s2 = 0x1p-900, up = 0x1p500, down = 0x1p-500, big = 0x1p900, dn = 0x1p-500, undn = 0x1p500; };
__device__ __forceinline__ float ba_qr_rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double ba_qr_rcp(double x) { return 1.0 / x; }

#ifdef BA_QR_STAMP2 /* dev tool (scripts/bench_qr.hip): where one wave's step goes -- stamps inside form / update_from of ONE (step, wave) */
#define BA_QR_FINE(i)                                                                                                                   \
    do {                                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                              \
        if (fine_now) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) ba_qr_fine[i] = (long long)t_; } \
        __builtin_amdgcn_sched_barrier(0);                                                                                              \
    } while (0)
#else
#define BA_QR_FINE(i) do { } while (0)
#endif
#define BA_QR_CWV 8 /* waves of a k_qr_chunk workgroup (two per SIMD: one's reductions and LDS round trips hide behind the other's FMAs) */
template <typename T, int NSB>
__global__ __launch_bounds__(64 * BA_QR_CWV) void k_qr_chunk(T *__restrict__ A, size_t lda, int c0, int bw, int row0, int level, long long stride, int nsb_total,
                                                  T *__restrict__ Tout /* [chunks][32 * 32] */, int nch,
                                                  const int *__restrict__ go = nullptr /* device-side LM control: *go == 0 -> nothing to do (MOREQR's outer QR) */)
{
    if (go && *go == 0) return; // (uniform)
    constexpr int NW = BA_QR_CWV, CH = BA_QR_PB * NSB, RPL = CH / 64, CW = BA_QR_PB / NW, RTW = CH / (16 * NW);
    static_assert(RTW >= 1, "a wave owns at least one row tile of the Gram product");
    __shared__ T vs[2][CH];
    __shared__ T tj_s[2], taus[BA_QR_PB];
    __shared__ T Gp[NW][BA_QR_PB][BA_QR_PB + 1], Gs[BA_QR_PB][BA_QR_PB + 1]; // partial Gram matrices of the waves; G, column j at Gs[j][.]
    __shared__ __attribute__((aligned(16))) T St[NW][16 * BA_QR_TP]; // per wave: a 16 x 32 tile of V between the load layout and the operand layout
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = blockIdx.x;
    const int nsb = min(NSB, nsb_total - g * NSB), rows = BA_QR_PB * nsb;
    if (threadIdx.x < BA_QR_PB) taus[threadIdx.x] = (T)0;
    const int hw_sqrt = ba_qr_hw_sqrt_flag, dbg_bits = ba_qr_dbg_flag; // (read ONCE: a load of a global per reflector step sits on the panel's critical path)
#ifdef BA_QR_STAMP
    if (nch == 1 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ba_qr_stamp[36] = (long long)t_; }
#endif
    // The chunk's columns live in PAIRS of rows per register pair (rows lane + 64 e, e = 2 k and 2 k + 1): the dot products and the
    // rank-1 updates of the step loop are v_pk_fma_f32 in fp32 -- two rows per instruction, the packed rate is the fp32 vector peak --
    // (fp64: two v_fma_f64 per pair, nothing lost).  The step loop is bound by the issue rate of the vector unit (two waves per SIMD).
    typedef T T2 __attribute__((ext_vector_type(2)));
    static_assert(RPL % 2 == 0, "rows per lane come in pairs");
    constexpr int RP2 = RPL / 2;
    T2 a[RP2][CW];
    size_t grow[RPL];
#pragma unroll
    for (int e = 0; e < RPL; e++) {
        const int l = lane + 64 * e;
        grow[e] = ba_qr_row<NSB>(row0, g, l < rows ? l : 0, stride);
#pragma unroll
        for (int q = 0; q < CW; q++) {
            const int c = NW * q + wv;
            a[e >> 1][q][e & 1] = A[(size_t)(c0 + (c < bw ? c : 0)) * lda + grow[e]]; // (unconditional; masked below, behind ALL the loads)
        }
    }
#pragma unroll
    for (int e = 0; e < RPL; e++) {
        const int l = lane + 64 * e;
#pragma unroll
        for (int q = 0; q < CW; q++) {
            const int c = NW * q + wv;
            T x = a[e >> 1][q][e & 1];
            asm volatile("" : "+v"(x));
            a[e >> 1][q][e & 1] = (c < bw && l < rows && (level == 1 || (l & 31) <= c)) ? x : (T)0;
        }
    }
    // reflector of column j from this wave's register column `pos` (a compile-time position after unrolling): into the LDS buffer
    // j & 1, the column retired to memory.  (Rows l = lane + 64 e with e >= 1 lie below every pivot: j < 32.)
#ifdef BA_QR_STAMP2
    bool fine_now = false;
#endif
    T keep[RPL]; // the owner's retired column between form (in front of the step's barrier) and retire (behind it)
    auto form = [&](int pos, int j) {
        BA_QR_FINE(0);
        const T a0 = a[0][pos][0]; // row `lane`: the only one that can be the pivot row or lie above it
        T2 p2 = {lane > j ? a0 * a0 : (T)0, a[0][pos][1] * a[0][pos][1]};
#pragma unroll
        for (int k = 1; k < RP2; k++) p2 = __builtin_elementwise_fma(a[k][pos], a[k][pos], p2);
        BA_QR_FINE(1);
        const T x2 = ba_wave_sum_all<T>(p2[0] + p2[1]);
        BA_QR_FINE(2);
        const T alpha = ba_readlane_dyn(a0, j); // row j lives in lane j, e = 0 (j < 32); j is wave-uniform: v_readlane, no LDS trip
        T tj = 0, sc = 0, beta = alpha;
        if (x2 != (T)0) { // (a column that is already zero below its pivot keeps the identity reflector)
            T s2 = alpha * alpha + x2, al = alpha, up = (T)1;
            // A column whose entries are so small that their SQUARES fall into the denormal range (fp64: entries below ~1e-150, the
            // noise-of-noise a chunk of low row rank leaves in its later columns; fp32: real data, a k1 / k2 column with entries of
            // ~1e-21): the sum of squares above has a few bits, beta and tau stop fitting v, and H = I - tau v v^T is not orthogonal
            // any more (round 4's self-check: 5e-2 off in one chunk of problem-21's panel 2; profiles/r04_qr_selfcheck.txt).  LAPACK's
            // dlarfg rescales such a column; so does this branch (rare, wave-uniform): the norm again from entries scaled by a power
            // of two, v = (x up) / (alpha up - beta up), beta = (beta up) / up.  A column that is zero even then keeps the identity.
            // The same branch, scaling DOWN, takes a squared norm that is huge or has overflowed (fp32: outlier observations with entries of
            // 1e19 and more; rounds 2 - 3 answered those with a NaN panel, i.e. a rejected trial).
            T unup = (T)1; // 1 / up
            if (s2 < ba_qr_tiny<T>::s2 || !(s2 < ba_qr_tiny<T>::big)) {
                const bool small = s2 < ba_qr_tiny<T>::s2;
                up = small ? ba_qr_tiny<T>::up : ba_qr_tiny<T>::dn;
                unup = small ? ba_qr_tiny<T>::down : ba_qr_tiny<T>::undn;
                const T y0 = a0 * up, y1 = a[0][pos][1] * up;
                T ps = (lane > j ? y0 * y0 : (T)0) + y1 * y1;
#pragma unroll
                for (int k = 1; k < RP2; k++) {
                    const T2 y = a[k][pos] * up;
                    ps += y[0] * y[0] + y[1] * y[1];
                }
                al = alpha * up;
                const T x2s = ba_wave_sum_all<T>(ps);
                s2 = x2s != (T)0 ? al * al + x2s : (T)0;
            }
            const T nb = s2 != (T)0 ? ba_qr_sqrt(s2, hw_sqrt) : (T)0;
            // nb == 0 with s2 != 0: the bare v_sqrt_f32 (BA_QR_HW_SQRT=1) flushes a denormal argument -- round 3's "no LM step accepted"
            // at config 3 with the hardware square root: beta = 0, tau = 0 * inf = NaN (profiles/EXPERIMENTS.md 6.3).  Identity reflector.
            if (nb != (T)0 && nb < (T)__builtin_inff()) { // (a norm that is zero or not finite even after rescaling leaves the identity)
                const T bs = al > (T)0 ? -nb : nb;
                const T tjn = (bs - al) * ba_qr_rcp(bs), scn = up * ba_qr_rcp(al - bs);
                if (scn < (T)__builtin_inff() && scn > -(T)__builtin_inff()) { // (|alpha| + |x| a denormal: 1 / it overflows -- a zero column to working precision)
                    tj = tjn;
                    sc = scn;
                    beta = bs * unup;
                }
            }
        }
        BA_QR_FINE(3);
        if (lane == 0) { taus[j] = tj; tj_s[j & 1] = tj; }
        // the hand-over (the others wait for it); the column's way to memory follows BEHIND the step's barrier (retire): its eight
        // stores with their address selects are 380 cycles that nobody has to wait for (in-kernel stamps, round 4)
        {
            const T ve = lane > j ? a0 * sc : (lane == j ? (T)1 : (T)0);
            vs[j & 1][lane] = ve;
            keep[0] = lane > j ? ve : (lane == j ? beta : a0);
        }
#pragma unroll
        for (int e = 1; e < RPL; e++) {
            const T ve = a[e >> 1][pos][e & 1] * sc;
            vs[j & 1][lane + 64 * e] = ve;
            keep[e] = ve;
        }
        BA_QR_FINE(4);
    };
    // column j to memory (R entries above the pivot, beta on it, v below): a masked-out element goes to a scratch word (this chunk's T
    // block, which the tail overwrites behind a full barrier) -- a select on the address instead of a branch per element
    auto retire = [&](int j) {
        T *const junk = Tout + (size_t)g * (BA_QR_PB * BA_QR_PB) + lane;
#pragma unroll
        for (int e = 0; e < RPL; e++) {
            const int l = lane + 64 * e;
            T *dst = (l < rows && (level == 1 || (l & 31) <= j)) ? A + (size_t)(c0 + j) * lda + grow[e] : junk;
            *dst = keep[e];
        }
        BA_QR_FINE(5);
    };
    // w_c = tau (v . a_c), a_c -= v w_c for the register columns q0 .. CW - 1 (v is 1 on its pivot row, zero above): all dot products
    // first, then all wave reductions, then the updates -- the DPP chains of the columns interleave
    auto update_from = [&](int q0, const T2 (&v)[RP2], T tj) {
        T pd[CW];
#pragma unroll
        for (int q = q0; q < CW; q++) {
            T2 d2 = v[0] * a[0][q];
#pragma unroll
            for (int k = 1; k < RP2; k++) d2 = __builtin_elementwise_fma(v[k], a[k][q], d2);
            pd[q] = d2[0] + d2[1];
        }
        BA_QR_FINE(8);
        static_assert(CW == 4, "ba_wave_sum4_all reduces the four register columns of a wave");
        {
            T pv[4];
#pragma unroll
            for (int q = 0; q < CW; q++) pv[q] = q >= q0 ? pd[q] : (T)0;
            if (dbg_bits & 4) { // (diagnostic: one reduction per column, as before round 4)
#pragma unroll
                for (int q = 0; q < CW; q++) pv[q] = ba_wave_sum_all<T>(pv[q]);
            } else
                ba_wave_sum4_all<T>(pv, lane); // the dot products of all live columns in one batched wave reduction
#pragma unroll
            for (int q = q0; q < CW; q++) pd[q] = tj * pv[q];
        }
        BA_QR_FINE(9);
#pragma unroll
        for (int q = q0; q < CW; q++) {
            const T2 npd = {-pd[q], -pd[q]};
#pragma unroll
            for (int k = 0; k < RP2; k++) a[k][q] = __builtin_elementwise_fma(v[k], npd, a[k][q]);
        }
        BA_QR_FINE(10);
    };
    // In-kernel stamps (scripts/bench_qr.hip -DBA_QR_STAMP, one-workgroup launch): 1.5 - 2.0 k cycles per step at 512 rows, of which a
    // wave's update of its 3 - 4 live columns is 900 - 1400 (two waves share a SIMD's vector unit; 27 dependent-ish instructions per
    // column: 8 FMAs, an 11-instruction DPP reduction, 8 FMAs) and the owner's reflector ~700.  Measured and not kept, all within 2 %:
    // the next owner updating its column first and OWING the rest until behind the next barrier (so that the reflector forms beside
    // the others' updates), v_readlane instead of the shuffle for the pivot, the hardware square root (which broke config 3).
    __syncthreads();
#pragma unroll
    for (int jq = 0; jq < CW; jq++) {
#pragma unroll 1
        for (int jw = 0; jw < NW; jw++) {
            const int j = NW * jq + jw;
            if (j >= bw) break;            // (uniform; only the last panel is narrower than 32)
#ifdef BA_QR_STAMP2
            fine_now = nch == 1 && j == 8 && (wv == 0 || wv == 5); // wave 0 owns column 8 (form), wave 5 updates three columns
#endif
            if (jw == wv) { form(jq, j); if (dbg_bits & 8) retire(j); } // (wave-uniform) this wave's column jq is column j
            // LDS-only barrier: the hand-over goes through LDS; __syncthreads() would also wait for the owner's global stores of the
            // retired column
            if (dbg_bits & 1) __syncthreads();
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef BA_QR_STAMP
            long long t_exit = 0;
            if (nch == 1) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); t_exit = (long long)t_; if (threadIdx.x == 0) ba_qr_stamp[j] = t_exit; }
#endif
            if (jw == wv && !(dbg_bits & 8)) retire(j);
#ifdef BA_QR_STAMP2
            if (wv != 5) fine_now = false;
#endif
            BA_QR_FINE(6);
            const T tj = tj_s[j & 1];
            T2 v[RP2];
#pragma unroll
            for (int k = 0; k < RP2; k++) { v[k][0] = vs[j & 1][lane + 128 * k]; v[k][1] = vs[j & 1][lane + 128 * k + 64]; }
            BA_QR_FINE(7);
            if (wv > jw) update_from(jq, v, tj);              // column NW jq + wv > j: register columns jq .. CW - 1 are live
            else if (jq + 1 < CW) update_from(jq + 1, v, tj); // column jq of this wave is retired (or being retired)
#ifdef BA_QR_STAMP
            if (nch == 1 && lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ba_qr_busy[8 * j + wv] = (long long)t_ - t_exit; }
#endif
        }
    }
#ifdef BA_QR_STAMP
    if (nch == 1 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ba_qr_stamp[32] = (long long)t_; }
#endif
    // ---- T factor.  The retired columns are in memory (written by different waves of this workgroup: visible behind the barrier).
    __syncthreads();
    if (dbg_bits & 2) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }
    {
        typedef typename ba_acc<T>::type acc_t;
        const int i = lane & 15, q = lane >> 4;
        acc_t g00, g01, g11;
#pragma unroll
        for (int v = 0; v < 4; v++) { g00[v] = 0; g01[v] = 0; g11[v] = 0; }
        // V by 64-byte runs of rows (lane (i, q): row i of the columns 4 s + q), into the operand order (lane <-> column, K-step <->
        // row) through this wave's LDS tile -- like k_qr_apply
        T *st = St[wv];
        constexpr int RB = RTW < 4 ? RTW : 4; // row tiles whose loads are in flight together
#pragma unroll 1
        for (int rb = 0; rb < RTW; rb += RB) {
            T vl[RB][8];
#pragma unroll
            for (int rt = 0; rt < RB; rt++) {
                const int l = (wv * RTW + rb + rt) * 16 + i;
                const size_t gl = ba_qr_row<NSB>(row0, g, l < rows ? l : 0, stride);
#pragma unroll
                for (int t = 0; t < 8; t++) vl[rt][t] = ba_qr_vraw<T>(A, lda, c0, bw, gl, 8 * q + t);
            }
#pragma unroll
            for (int rt = 0; rt < RB; rt++) {
                const int l = (wv * RTW + rb + rt) * 16 + i;
#pragma unroll
                for (int t = 0; t < 8; t++) vl[rt][t] = ba_qr_vmask<T>(vl[rt][t], bw, level, rows, l, 8 * q + t);
            }
#pragma unroll
            for (int rt = 0; rt < RB; rt++) {
                ba_qr_tile_to_lds<T>(st, vl[rt], i, q);
                ba_wave_lds_order();
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    // G[i][j] += V[l][i] V[l][j]: lane (i, q) holds V[row crow(q, v)][i], which is both the A and the B fragment
                    const T v0 = st[ba_crow<T>(q, v) * BA_QR_TP + i], v1 = st[ba_crow<T>(q, v) * BA_QR_TP + 16 + i];
                    g00 = ba_mfma(v0, v0, g00);
                    g01 = ba_mfma(v0, v1, g01);
                    g11 = ba_mfma(v1, v1, g11);
                }
                ba_wave_lds_order();
            }
        }
#pragma unroll
        for (int v = 0; v < 4; v++) { // C/D fragment: row = crow(q, v), column = i
            const int r = ba_crow<T>(q, v);
            Gp[wv][r][i] = g00[v];
            Gp[wv][r][16 + i] = g01[v];
            Gp[wv][16 + r][16 + i] = g11[v];
            Gp[wv][16 + r][i] = (T)0;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BA_QR_PB * BA_QR_PB; idx += 64 * NW) {
        const int k = idx >> 5, j = idx & 31;
        T sum = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) sum += Gp[w][k][j];
        Gs[j][k] = sum;
    }
    __syncthreads();
    if (wv == 0) {
        T Trow[BA_QR_PB]; // row `lane` of T (lanes 32 .. 63 idle along)
#pragma unroll
        for (int k = 0; k < BA_QR_PB; k++) Trow[k] = (T)0;
#pragma unroll
        for (int j = 0; j < BA_QR_PB; j++) {
            const T tj = taus[j];
            T acc = 0;
#pragma unroll
            for (int k = 0; k < j; k++) acc += Trow[k] * Gs[j][k]; // (Trow[k] = 0 for k < lane: T is upper triangular)
            Trow[j] = lane == j ? tj : (lane < j ? -tj * acc : (T)0);
        }
        if (lane < BA_QR_PB) {
            T *to = Tout + (size_t)g * (BA_QR_PB * BA_QR_PB) + (size_t)lane * BA_QR_PB;
#pragma unroll
            for (int k = 0; k < BA_QR_PB; k++) to[k] = Trow[k];
        }
#ifdef BA_QR_STAMP
        if (nch == 1 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ba_qr_stamp[33] = (long long)t_; }
#endif
    }
}

// ---- the reflectors of one chunk applied to 32 trailing columns, on the matrix cores -------------------------------------------------
// One workgroup of EIGHT waves per (chunk, strip of 32 trailing columns): Q^T B = (I - V T^T V^T) B with the chunk's V (CH x 32) and T
// (k_qr_chunk), as three products of the 16x16x4 instruction.  The launch is bound by the bytes it moves (~4 TB/s of 64-byte runs),
// so the strip and the chunk's V are read from memory ONCE: wave w keeps its eighth of the rows in registers from the first load to
// the last store -- B as C/D fragments (CH / 128 x 2 tiles = 64 registers), V in the order it is loaded in (64 registers):
//   W = V^T B   the fragment registers of B ARE the B operands (register v of a tile = the rows crow(q, v) of its four K-steps), V is
//               brought into the same row order; the eight waves' partial sums meet in LDS (fixed order);
//   Z = T^T W   one 16 x 16 tile on each of four waves;
//   B -= V Z    eight K-steps per tile: step s takes the reflectors 8 q + s (lane (m, q) holds V[m][8 q + s] from the load; any
//               one-to-one map of (s, q) onto the 32 reflectors does, as long as Z is read through the same map).
// Every global access is a 64-byte run of rows per column (lane <-> row; a fragment loaded lane <-> column puts every lane of an
// instruction into a cache line of its own and the L2 moves sixteen times the tile); the fragments' lane <-> column order is reached
// through a per-wave LDS tile (ba_qr_tile_*; no barrier: a wave's LDS instructions execute in order).  All loads of a batch are
// issued before the first is looked at (ba_qr_vraw / ba_qr_vmask).
// Measured at config 3, first panel's launch (1840 tasks): 180 us = 3.9 TB/s.  A streaming form (nothing held between the passes,
// five workgroups per CU, next tile prefetched) reads B and V twice and runs at the same ~4.3 TB/s: 250 - 270 us.
// bw < 32 (last panel): the missing reflectors are zero columns of V and T.  Rows past the matrix (last chunk) and columns past the
// strip are masked.  Workgroup -> task: the strips of one chunk run next to each other on ONE XCD (blockIdx % 8 under the observed
// round-robin placement).
#define BA_QR_AW 8 /* waves of a k_qr_apply workgroup */
template <typename T, int NSB>
__global__ __launch_bounds__(64 * BA_QR_AW) void k_qr_apply(T *__restrict__ A, size_t lda, int c0, int bw, int row0, int level, long long stride, int nsb_total,
                                                           const T *__restrict__ Tg, int col0, int col1, int nch, int nct,
                                                           const int *__restrict__ go = nullptr)
{
    if (go && *go == 0) return; // (uniform)
    constexpr int CH = BA_QR_PB * NSB, RTW = CH / (16 * BA_QR_AW), PB = BA_QR_PB;
    static_assert(RTW >= 1 && PB == 32, "a wave owns at least one row tile; the tile helpers are written for 32 columns");
    typedef typename ba_acc<T>::type acc_t;
    __shared__ T Wp[BA_QR_AW][PB][PB + 1], Zs[PB][PB + 1];
    __shared__ __attribute__((aligned(16))) T St[BA_QR_AW][16 * BA_QR_TP]; // per wave: one 16 x 32 tile between the two layouts, [row][column]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = 8 * (slot / nct) + xcd, ct = slot % nct;
    if (g >= nch) return; // (uniform)
    const int cb = col0 + PB * ct, ncol = min(PB, col1 - cb);
    const int nsb = min(NSB, nsb_total - g * NSB), rows = PB * nsb;
    T *st = St[wv];
    // ---- all of V for this wave's rows first (lane (i, q): row i of the reflectors 8 q .. 8 q + 7; kept to the end)
    T vm[RTW][8];
    size_t grl[RTW]; // global row of this lane's row of tile rt (clamped)
#pragma unroll
    for (int rt = 0; rt < RTW; rt++) {
        const int l = (wv * RTW + rt) * 16 + i;
        grl[rt] = ba_qr_row<NSB>(row0, g, l < rows ? l : 0, stride);
#pragma unroll
        for (int t = 0; t < 8; t++) vm[rt][t] = ba_qr_vraw<T>(A, lda, c0, bw, grl[rt], 8 * q + t);
    }
    acc_t Bt[RTW][2], Wt[2][2];
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int jt = 0; jt < 2; jt++)
#pragma unroll
            for (int v = 0; v < 4; v++) Wt[it][jt][v] = 0;
    // ---- W = V^T B (this wave's rows), the strip on its way into the C/D fragments
    constexpr int RB = RTW < 4 ? RTW : 4; // row tiles whose loads of B are in flight together
#pragma unroll
    for (int rb = 0; rb < RTW; rb += RB) {
        T bl[RB][8];
#pragma unroll
        for (int rt = 0; rt < RB; rt++)
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int c = 8 * q + t;
                bl[rt][t] = A[(size_t)(cb + (c < ncol ? c : 0)) * lda + grl[rb + rt]];
            }
        if (rb == 0) { // the masks of V behind the first batch of B's loads: everything the wave reads is in flight by now
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) {
                const int l = (wv * RTW + rt) * 16 + i;
#pragma unroll
                for (int t = 0; t < 8; t++) vm[rt][t] = ba_qr_vmask<T>(vm[rt][t], bw, level, rows, l, 8 * q + t);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RB; rt++) {
            const int l = (wv * RTW + rb + rt) * 16 + i;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                T x = bl[rt][t];
                asm volatile("" : "+v"(x));
                bl[rt][t] = (l < rows && 8 * q + t < ncol) ? x : (T)0;
            }
            ba_qr_tile_to_lds<T>(st, bl[rt], i, q);
            ba_wave_lds_order();
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int v = 0; v < 4; v++) Bt[rb + rt][jt][v] = st[ba_crow<T>(q, v) * BA_QR_TP + 16 * jt + i];
            ba_wave_lds_order();
            ba_qr_tile_to_lds<T>(st, vm[rb + rt], i, q);
            ba_wave_lds_order();
#pragma unroll
            for (int v = 0; v < 4; v++) { // (rows past the chunk: V and B are zero there)
                const T v0 = st[ba_crow<T>(q, v) * BA_QR_TP + i], v1 = st[ba_crow<T>(q, v) * BA_QR_TP + 16 + i];
                Wt[0][0] = ba_mfma(v0, Bt[rb + rt][0][v], Wt[0][0]);
                Wt[0][1] = ba_mfma(v0, Bt[rb + rt][1][v], Wt[0][1]);
                Wt[1][0] = ba_mfma(v1, Bt[rb + rt][0][v], Wt[1][0]);
                Wt[1][1] = ba_mfma(v1, Bt[rb + rt][1][v], Wt[1][1]);
            }
            ba_wave_lds_order();
        }
    }
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int jt = 0; jt < 2; jt++)
#pragma unroll
            for (int v = 0; v < 4; v++) Wp[wv][16 * it + ba_crow<T>(q, v)][16 * jt + i] = Wt[it][jt][v];
    __syncthreads();
    // ---- Z = T^T W: waves 0 .. 3 form the tile (w >> 1, w & 1)
    if (wv < 4) {
        const int it = wv >> 1, jt = wv & 1;
        const T *Tc = Tg + (size_t)g * (PB * PB);
        acc_t z;
#pragma unroll
        for (int v = 0; v < 4; v++) z[v] = 0;
#pragma unroll
        for (int s = 0; s < PB / 4; s++) {
            const int k = 4 * s + q;
            const T ta = Tc[(size_t)k * PB + 16 * it + i]; // A[i][k] = T[k][i]
            T wb = 0;
#pragma unroll
            for (int w = 0; w < BA_QR_AW; w++) wb += Wp[w][k][16 * jt + i];
            z = ba_mfma(ta, wb, z);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) Zs[16 * it + ba_crow<T>(q, v)][16 * jt + i] = z[v];
    }
    __syncthreads();
    // ---- B -= V Z, and back to memory the way it came
    T zb[8][2];
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int jt = 0; jt < 2; jt++) zb[s][jt] = Zs[8 * q + s][16 * jt + i]; // B[k][n] = Z[8 q + s][n]
#pragma unroll
    for (int rt = 0; rt < RTW; rt++) {
        const int lt = (wv * RTW + rt) * 16, l = lt + i;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            Bt[rt][0] = ba_mfma(-vm[rt][s], zb[s][0], Bt[rt][0]);
            Bt[rt][1] = ba_mfma(-vm[rt][s], zb[s][1], Bt[rt][1]);
        }
        if (lt < rows) { // (wave-uniform)
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int v = 0; v < 4; v++) st[ba_crow<T>(q, v) * BA_QR_TP + 16 * jt + i] = Bt[rt][jt][v];
            ba_wave_lds_order();
            T o[8];
            ba_qr_tile_from_lds<T>(st, o, i, q);
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int c = 8 * q + t;
                if (l < rows && c < ncol) A[(size_t)(cb + c) * lda + grl[rt]] = o[t];
            }
            ba_wave_lds_order();
        }
    }
}

// ---- R y = (Q^T rhs)[0 : D): back substitution by one workgroup ------------------------------------------------------------
// R is the upper triangle of the first D rows of A, the transformed right-hand side column D.  Blocks of 64 unknowns from the bottom:
// wave 0 solves the 64 x 64 triangle (lane i = row i, the unknowns handed down by readlane: no barrier inside a block), then all
// 256 threads eliminate the block from the rows above it (thread = row, the block's columns of R walked down their contiguous
// direction).  Six blocks at D = 351 instead of 351 steps with two barriers each (0.24 ms -> ~0.03 ms).
template <typename T>
__global__ __launch_bounds__(256) void k_qr_backsolve(const T *__restrict__ A, size_t lda, int D, T *__restrict__ y)
{
    extern __shared__ unsigned char smem_raw[];
    T *b = reinterpret_cast<T *>(smem_raw); // D right-hand sides, 64 unknowns of the current block, its 64 x 64 triangle [k][i]
    T *yb = b + D, *tri = yb + 64;
    const int tid = threadIdx.x;
    for (int i = tid; i < D; i += 256) b[i] = A[(size_t)D * lda + i];
    for (int j1 = D; j1 > 0; j1 -= 64) {
        const int j0 = j1 > 64 ? j1 - 64 : 0, nb = j1 - j0;
        for (int idx = tid; idx < 64 * 64; idx += 256) { // the block's triangle, by runs of rows (a load inside the solve loop would put an
            const int k = idx >> 6, i = idx & 63;        // L2 round trip into every one of its 64 dependent steps)
            tri[idx] = (k < nb && i <= k) ? A[(size_t)(j0 + k) * lda + j0 + i] : (T)0;
        }
        __syncthreads();
        if (tid < 64) { // wave 0: rows j0 .. j1 - 1, lane i holds b[j0 + i]
            const int i = tid;
            T bi = i < nb ? b[j0 + i] : (T)0;
            const T dii = i < nb ? tri[64 * i + i] : (T)1;
            for (int k = nb - 1; k >= 0; k--) {
                const T yk = ba_readlane_dyn(bi, k) / ba_readlane_dyn(dii, k);
                bi -= (i < k ? tri[64 * k + i] : (T)0) * yk;
                if (i == k) { yb[k] = yk; y[j0 + k] = yk; }
            }
        }
        __syncthreads();
        for (int i = tid; i < j0; i += 256) { // rows above the block (eight loads in flight: one per trip of the loop paid the L2 latency 64 times)
            T acc = b[i];
            int k = 0;
            for (; k + 8 <= nb; k += 8) {
                T r8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) r8[u] = A[(size_t)(j0 + k + u) * lda + i];
#pragma unroll
                for (int u = 0; u < 8; u++) acc -= r8[u] * yb[k + u];
            }
            for (; k < nb; k++) acc -= A[(size_t)(j0 + k) * lda + i] * yb[k];
            b[i] = acc;
        }
        __syncthreads();
    }
}

// Host side: QR of the (mrows x D) matrix A (+ rhs in column D) on `st`, then y = argmin || A y - rhs ||.
// A: lda >= mrows + 64 rows allocated and zero beyond mrows.  tau: the T factors, room for (ceil(mrows / CH) + 2) * 32 * 32 scalars per
// level, 8 levels (CH = 1024 / 512 rows: 181 633 rows are 3 levels in fp32, 4 in fp64).
// st2 != nullptr (with two events): the trailing updates run on st2 beside the panel's chunk chain -- level L + 1 of the chain only
// needs the panel's own R's from level L, not the trailing update of level L -- and the next panel waits for the last of them
// (fork / join by events: also valid inside a stream capture).  6.2 -> 5.3 ms per trial at config 3 in round 2.  Look-ahead (every
// level's reflectors to the next panel's 32 columns first, so that the next chain starts before the rest is done): on `st` itself it
// cost more than it gave in round 2 (5.7 ms: five more launches on the critical stream); round 3 puts those launches on a THIRD
// stream -- the chain stream carries nothing but the chunk kernels (ba_qr_side).
// ---- sharded QRKIT: distributed TSQR ----------------------------------------------------------------------------------------------
// Every shard factors the rows of J2bot it owns (ba_qr_factor); what it contributes to the whole matrix's factor is its D x D
// triangle R_r and the head of Q_r^T rhs.  k_qr_stack_pack copies both into block r of a zeroed (world D) x (D + 1) matrix, behind
// it this shard's camera gradient g_c (D) and its part of the energy (1): ONE sum all-reduce of that buffer gives every shard the
// stack [R_0; R_1; ...] (+ rhs), the global g_c and the energy; the QR of the stack (same kernels, ~world D rows) and the back
// substitution then run redundantly.  No normal equations anywhere: cond stays that of J2bot.
template <typename T>
__global__ __launch_bounds__(256) void k_qr_stack_pack(const T *__restrict__ A, size_t lda, int D, int rank, T *__restrict__ B, size_t ldb, size_t nmat,
                                                       const T *__restrict__ gc, const T *__restrict__ scal, int eloc)
{
    const int c = blockIdx.x; // column 0 .. D (D = the right-hand side)
    if (c > D) { // the tail: g_c, energy
        for (int i = threadIdx.x; i < D; i += 256) B[nmat + i] = gc[i];
        if (threadIdx.x == 0) B[nmat + D] = scal[eloc];
        return;
    }
    const int top = c < D ? c : D - 1; // rows 0 .. top of the column belong to R (the reflectors sit below)
    for (int i = threadIdx.x; i <= top; i += 256) B[(size_t)c * ldb + (size_t)rank * D + i] = A[(size_t)c * lda + i];
}
template <typename T>
__global__ __launch_bounds__(256) void k_qr_stack_unpack(const T *__restrict__ B, size_t nmat, int D, T *__restrict__ gc_out, T *__restrict__ scal, int etot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < D) gc_out[i] = B[nmat + i];
    if (i == 0) scal[etot] = B[nmat + D];
}

template <typename T> inline void ba_qr_backsolve(hipStream_t st, const T *A, size_t lda, int D, T *y)
{
    hipLaunchKernelGGL((k_qr_backsolve<T>), dim3(1), dim3(256), sizeof(T) * (size_t)(D + 64 + 64 * 64), st, A, lda, D, y);
}

// Side streams of the factorisation (all nullptr: everything on `st`).  st2 alone: the trailing updates beside the chunk chain.
// st2 + st3 (look-ahead): a level's reflectors go to the NEXT panel's 32 columns on st3 and to the rest on st2; the next chain
// waits for st3 only.
#define BA_QR_TAU_LEVELS 8 /* TSQR levels the T storage has room for (16^8 chunks) */
struct ba_qr_side {
    hipStream_t st2 = nullptr, st3 = nullptr;
    hipEvent_t ev_chunk = nullptr, ev_apply = nullptr, ev_next = nullptr; // chunk done (st) | rest updated (st2) | next panel updated (st3)
    const int *go = nullptr; // every kernel of the factorisation returns at once while *go == 0 (MOREQR's outer QR behind a rejected trial)
    bool two() const { return st2 && ev_chunk && ev_apply; }
    bool lookahead() const { return two() && st3 && ev_next; }
};

template <typename T>
inline void ba_qr_factor(hipStream_t st, T *A, size_t lda, int mrows, int D, T *tau_all, size_t tau_level_stride, const ba_qr_side &sd = ba_qr_side())
{
    // tau_all: BA_QR_TAU_LEVELS level slots of tau_level_stride scalars, TWICE with look-ahead: the chain of panel p + 1 writes its T
    // factors while st2 still applies panel p's
    constexpr int NSB1 = ba_qr_cfg<T>::NSB, NSBU = ba_qr_cfg<T>::NSBU;
    const bool two = sd.two(), la = sd.lookahead();
    bool rest_pending = false, next_pending = false; // st2 / st3 hold updates that `st` (or the other one) has not waited for yet
    auto apply = [&](hipStream_t sa, int level, int nch, int nsb, long long stride, const T *tl, int c0, int bw, int col0, int col1) {
        const int nct = (col1 - col0 + BA_QR_PB - 1) / BA_QR_PB; // strips of 32 columns
        // (8 x ceil(nch / 8) x nct workgroups: chunk g's strips sit at blockIdx % 8 == g % 8)
        const dim3 ga((unsigned)(8ll * ((nch + 7) / 8) * nct));
        if (level == 1)
            hipLaunchKernelGGL((k_qr_apply<T, NSB1>), ga, dim3(64 * BA_QR_AW), 0, sa, A, lda, c0, bw, c0, level, stride, nsb, tl, col0, col1, nch, nct, sd.go);
        else
            hipLaunchKernelGGL((k_qr_apply<T, NSBU>), ga, dim3(64 * BA_QR_AW), 0, sa, A, lda, c0, bw, c0, level, stride, nsb, tl, col0, col1, nch, nct, sd.go);
    };
    const char *stop_env = getenv("BA_QR_STOP_PANEL"); // diagnostic: leave the factorisation behind this panel (its T factors stay in tau_all)
    const int stop_panel = stop_env ? atoi(stop_env) : -1;
    for (int c0 = 0; c0 < D; c0 += BA_QR_PB) {
        if (stop_panel >= 0 && c0 / BA_QR_PB > stop_panel) break;
        const int bw = D - c0 < BA_QR_PB ? D - c0 : BA_QR_PB;
        const int col0 = c0 + bw, col1 = D + 1; // trailing columns incl. the right-hand side
        const int colm = la ? (col0 + BA_QR_PB < col1 ? col0 + BA_QR_PB : col1) : col0; // look-ahead: [col0, colm) is the next panel
        int nsb = (mrows - c0 + BA_QR_PB - 1) / BA_QR_PB; // 32-row blocks from the panel's first row down
        long long stride = BA_QR_PB;
        T *tau = tau_all + (la && ((c0 / BA_QR_PB) & 1) ? (size_t)BA_QR_TAU_LEVELS * tau_level_stride : 0);
        if (next_pending) { // this panel's columns carry every earlier reflector once st3 is done with them
            (void)hipStreamWaitEvent(st, sd.ev_next, 0);
            next_pending = false;
        }
        bool first_next = true;
        for (int level = 1;; level++) {
            const int fan = level == 1 ? NSB1 : NSBU;
            const int nch = (nsb + fan - 1) / fan;
            T *tl = tau + (size_t)(level - 1) * tau_level_stride;
            if (level == 1) hipLaunchKernelGGL((k_qr_chunk<T, NSB1>), dim3(nch), dim3(64 * BA_QR_CWV), 0, st, A, lda, c0, bw, c0, level, stride, nsb, tl, nch, sd.go);
            else hipLaunchKernelGGL((k_qr_chunk<T, NSBU>), dim3(nch), dim3(64 * BA_QR_CWV), 0, st, A, lda, c0, bw, c0, level, stride, nsb, tl, nch, sd.go);
            if (col0 < col1) {
                if (two) (void)hipEventRecord(sd.ev_chunk, st);
                if (la) {
                    (void)hipStreamWaitEvent(sd.st3, sd.ev_chunk, 0);
                    if (first_next && rest_pending) (void)hipStreamWaitEvent(sd.st3, sd.ev_apply, 0); // the previous panels' reflectors come first
                    first_next = false;
                    apply(sd.st3, level, nch, nsb, stride, (const T *)tl, c0, bw, col0, colm);
                }
                if (colm < col1) {
                    if (two) (void)hipStreamWaitEvent(sd.st2, sd.ev_chunk, 0);
                    apply(two ? sd.st2 : st, level, nch, nsb, stride, (const T *)tl, c0, bw, colm, col1);
                }
            }
            if (nch == 1) break;
            nsb = nch;
            stride *= fan;
        }
        if (two && colm < col1) {
            (void)hipEventRecord(sd.ev_apply, sd.st2);
            rest_pending = true;
            if (!la) { // the next panel (and the back substitution) read what the trailing updates wrote
                (void)hipStreamWaitEvent(st, sd.ev_apply, 0);
                rest_pending = false;
            }
        }
        if (la && col0 < col1) {
            (void)hipEventRecord(sd.ev_next, sd.st3);
            next_pending = true;
        }
    }
    if (next_pending) (void)hipStreamWaitEvent(st, sd.ev_next, 0);
    if (rest_pending) (void)hipStreamWaitEvent(st, sd.ev_apply, 0);
}

template <typename T>
inline void ba_qr_solve(hipStream_t st, T *A, size_t lda, int mrows, int D, T *tau, size_t tau_level_stride, T *y, const ba_qr_side &sd = ba_qr_side())
{
    ba_qr_factor<T>(st, A, lda, mrows, D, tau, tau_level_stride, sd);
    ba_qr_backsolve<T>(st, (const T *)A, lda, D, y);
}

#endif
