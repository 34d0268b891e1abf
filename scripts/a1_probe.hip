// Dev tool: what bounds the 16x16 diagonal-tile pivot loop (A1 of ba_panel_body) on gfx950?  Single wave, variants.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bundleadjustment_benchmarks_amd/csrc scripts/a1_probe.hip -o scripts/a1_probe.bin
#include "ba_dense.hip.h"
#include <cstdio>
#include <vector>
template <int K> __device__ __forceinline__ double dppb(double v)
{ // broadcast lane K of every row of 16 lanes (DPP row_newbcast), no LDS
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + K, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dppb_k(double v, int k)
{
    switch (k) {
    case 0: return dppb<0>(v); case 1: return dppb<1>(v); case 2: return dppb<2>(v); case 3: return dppb<3>(v);
    case 4: return dppb<4>(v); case 5: return dppb<5>(v); case 6: return dppb<6>(v); case 7: return dppb<7>(v);
    case 8: return dppb<8>(v); case 9: return dppb<9>(v); case 10: return dppb<10>(v); case 11: return dppb<11>(v);
    case 12: return dppb<12>(v); case 13: return dppb<13>(v); case 14: return dppb<14>(v); default: return dppb<15>(v);
    }
}
__device__ __forceinline__ double bperm(double v, int src_lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(4 * src_lane, lo);
    hi = __builtin_amdgcn_ds_bpermute(4 * src_lane, hi);
    return __hiloint2double(hi, lo);
}
// value of lane (i, kq) in every lane (i, *): two row swaps (gfx950 v_permlane16_swap / v_permlane32_swap), no LDS
__device__ __forceinline__ double xrow(double v, int kq)
{
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const unsigned tl = (kq & 1) ? a[1] : a[0], th = (kq & 1) ? b[1] : b[0];
    auto c = __builtin_amdgcn_permlane32_swap(tl, tl, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(th, th, false, false);
    return __hiloint2double((int)((kq & 2) ? d[1] : d[0]), (int)((kq & 2) ? c[1] : c[0]));
}
template <int V> __global__ __launch_bounds__(256) void probe(long long *out, double *sink, const double *tile, int reps)
{
    typedef double T;
    __shared__ T Ad[16][17], colx4[4][16], junkbuf[64];
    __shared__ int prog[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
    if (tid < 256) { for (int e = tid; e < 256; e += blockDim.x) Ad[e / 16][e % 16] = tile[e]; }
    __syncthreads();
    if (wv != 0) return;
    T acc = 0;
    T *const junk = junkbuf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; rep++) {
        T a[4], lprev = 0, wi[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { a[c] = tile[(4 * q + c) * 16 + i] + (T)rep * 1e-9; wi[c] = (4 * q + c == i) ? 1.0 : 0.0; }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int kq = k >> 2, kc = k & 3;
            T lraw, y[4], r;
            if (V == 15) { // V14 with the inverse (W = L^-1) carried by the SAME wave: the update of pivot k - 1 fills the wait for the
                           // exchange of pivot k (its multiplier is the wave's own register, row k - 1 of W a DPP row broadcast)
                colx4[q][i] = a[kc];
                asm volatile("" ::: "memory");
                lraw = colx4[kq][i];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                asm volatile("" ::: "memory");
                if (k > 0) {
                    *((q == ((k - 1) >> 2)) ? &Ad[k - 1][i] : junk) = lprev;
                    T wk[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) wk[c] = ba_rowbcast_k(wi[c], k - 1);
#pragma unroll
                    for (int c = 0; c < 4; c++) wi[c] -= lprev * wk[c];
                }
                asm volatile("" ::: "memory");
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                r = ba_rcp(dk);
                __builtin_amdgcn_sched_barrier(0);
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                lprev = l;
                continue;
            }
            if (V >= 13) { // V11 with the L / progress stores of pivot k - 1 issued BEHIND the exchange loads of pivot k
                colx4[q][i] = a[kc];
                asm volatile("" ::: "memory");
                lraw = colx4[kq][i];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                asm volatile("" ::: "memory");
                if (k > 0) {
                    *((q == ((k - 1) >> 2)) ? &Ad[k - 1][i] : junk) = lprev;
                    __hip_atomic_store(&prog[lane], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                asm volatile("" ::: "memory");
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                r = ba_rcp(dk);
                if (V == 14) __builtin_amdgcn_sched_barrier(0); // the reciprocal's Newton steps in front of the wait for the loads
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                lprev = l;
                continue;
            }
            if (V >= 11) { // the shipped LDS exchange, but ordered for the compiler only: the LDS executes a wave's instructions in
                           // order, so the loads need not wait for the stores in front of them (no s_waitcnt from a fence)
                colx4[q][i] = a[kc];
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                r = ba_rcp(dk);
                asm volatile("" ::: "memory");
                lraw = colx4[kq][i];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                asm volatile("" ::: "memory");
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                if (V == 11) {
                    *((q == kq) ? &Ad[k][i] : junk) = l;
                    __hip_atomic_store(&prog[lane], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                asm volatile("" ::: "memory");
                continue;
            }
            if (V >= 9) { // no LDS at all: column k by row swaps, row k by DPP; stores of the previous pivot issued first
                if (V == 9 && k > 0) {
                    *((q == ((k - 1) >> 2)) ? &Ad[k - 1][i] : junk) = lprev;
                    __hip_atomic_store(&prog[lane], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                r = ba_rcp(dk);
                lraw = xrow(a[kc], kq);
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = dppb_k(a[c], k);
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                lprev = l;
                continue;
            }
            if (V >= 7) { // as V5, ordered by hand: permute in flight, then the previous pivot's stores, reciprocal + DPP, then the use
                int blo = __builtin_amdgcn_ds_bpermute(4 * (16 * kq + i), __double2loint(a[kc]));
                int bhi = __builtin_amdgcn_ds_bpermute(4 * (16 * kq + i), __double2hiint(a[kc]));
                __builtin_amdgcn_sched_barrier(0);
                if (V == 7 && k > 0) {
                    *((q == ((k - 1) >> 2)) ? &Ad[k - 1][i] : junk) = lprev;
                    __hip_atomic_store(&prog[lane], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_sched_barrier(0);
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                r = ba_rcp(dk);
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = dppb_k(a[c], k);
                __builtin_amdgcn_sched_barrier(0);
                lraw = __hiloint2double(bhi, blo);
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                lprev = l;
                continue;
            }
            if (V >= 5) { // no LDS exchange: column k by ds_bpermute, row k (= column k, the tile is symmetric) by DPP
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                r = ba_rcp(dk);
                lraw = bperm(a[kc], 16 * kq + i);
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = dppb_k(a[c], k);
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                if (V == 5) {
                    *((q == kq) ? &Ad[k][i] : junk) = l;
                    __hip_atomic_store(&prog[lane], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                continue;
            }
            if (V != 3) colx4[q][i] = a[kc];
            if (V != 4) {
                const T dk = ba_readlane(a[kc], 16 * kq + k);
                if (V == 2) { T r0 = __builtin_amdgcn_rcp(dk); r = fma(fma(-dk, r0, 1.0), r0, r0); }
                else r = ba_rcp(dk);
            } else r = 0.2;
            ba_wave_lds_sync();
            if (V != 3) {
                lraw = colx4[kq][i];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
            } else {
                lraw = a[kc];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = a[c] * 0.5;
            }
            const T lm = (i > k) ? lraw : (T)0;
            const T l = lm * r;
#pragma unroll
            for (int c = 0; c < 4; c++) a[c] -= l * y[c];
            if (V == 0) {
                *((q == kq) ? &Ad[k][i] : junk) = l;
                __hip_atomic_store(&prog[lane], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ba_wave_lds_sync();
        }
#pragma unroll
        for (int c = 0; c < 4; c++) acc += a[c] + wi[c];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[lane] = acc;
    if (lane == 0) out[0] = (long long)(t1 - t0);
}
// Two-wave form as in ba_panel_body: wave 0 factors, wave WINV inverts behind it (sentinel hand-off through Lbuf).
template <int WINV, int MODE> __global__ __launch_bounds__(256) void probe2(long long *out, double *sink, const double *tile, int reps)
{
    typedef double T;
    __shared__ T colx4[4][16], junk2[4][64], Lbuf[16][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
    Lbuf[tid >> 4][tid & 15] = ba_sentinel<T>();
    __syncthreads();
    T acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv == 0 && MODE != 2) {
        T *lst[4];
#pragma unroll
        for (int g = 0; g < 4; g++) lst[g] = (q == g) ? &Lbuf[4 * g][i] : &junk2[0][lane];
        for (int rep = 0; rep < reps; rep++) {
            T a[4];
#pragma unroll
            for (int c = 0; c < 4; c++) a[c] = tile[(4 * q + c) * 16 + i] + (T)rep * 1e-9;
            T r = ba_rcp(ba_readlane(a[0], 0));
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int kq = k >> 2, kc = k & 3;
                colx4[q][i] = a[kc];
                ba_wave_lds_sync();
                const T lraw = colx4[kq][i];
                T y[4];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                __builtin_amdgcn_sched_barrier(0);
                T rnext = (T)0;
                if (k + 1 < 16) {
                    const T x = ba_readlane(a[kc], 16 * kq + k + 1);
                    const T a11 = ba_readlane(a[(k + 1) & 3], 16 * ((k + 1) >> 2) + k + 1);
                    const T lx = x * r;
                    rnext = ba_rcp(__builtin_fma(-lx, x, a11));
                }
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] = __builtin_fma(-l, y[c], a[c]);
                if (MODE == 0) {
                    // wait until wave WINV has consumed (re-armed) the previous repetition's column
                    if (rep > 0) while (__builtin_amdgcn_ballot_w64(!ba_is_sentinel(__hip_atomic_load(&Lbuf[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) != 0);
                    lst[kq][kc * 16] = l;
                }
                ba_wave_lds_sync();
                r = rnext;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) acc += a[c];
        }
    } else if (wv == WINV && MODE != 1) {
        const int j = i;
        for (int rep = 0; rep < reps; rep++) {
            T w[16];
#pragma unroll
            for (int r = 0; r < 16; r++) w[r] = (r == j) ? (T)1 : (T)0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                T lk_;
                if (MODE == 0) {
                    do lk_ = __hip_atomic_load(&Lbuf[k][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    while (__builtin_amdgcn_ballot_w64(ba_is_sentinel(lk_)) != 0);
                } else lk_ = tile[k * 16 + j] * 1e-3 + w[15] * 1e-9;
#pragma unroll
                for (int r = k + 1; r < 16; r++) w[r] = __builtin_fma(-ba_readlane(lk_, r), w[k], w[r]);
                if (q == 0 && MODE == 0) Lbuf[k][j] = ba_sentinel<T>();
            }
#pragma unroll
            for (int r = 0; r < 16; r++) acc += w[r];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[tid] = acc;
    if (lane == 0) out[wv] = (long long)(t1 - t0);
}
// What slows the factor wave inside the panel kernel?  Wave 0 runs the shipped pivot loop (V14); MODE bit 0: wave 1 follows it
// through the progress word like the inverse wave; bit 1: waves 2 and 3 issue f64 MFMAs back to back; bit 2: waves 2 and 3 read
// and write LDS tiles like the rank-16 updates.
template <int MODE> __global__ __launch_bounds__(256) void probe3(long long *out, double *sink, const double *tile, int reps)
{
    typedef double T;
    __shared__ T Ad[16][17], colx4[4][16], junkbuf[64], Big[64][65];
    __shared__ int prog[64], done;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, q = lane >> 4;
    for (int e = tid; e < 256; e += 256) Ad[e / 16][e % 16] = tile[e];
    for (int e = tid; e < 64 * 65; e += 256) Big[e / 65][e % 65] = 1e-3 * (e % 7);
    if (tid < 64) prog[tid] = 0;
    if (tid == 0) done = 0;
    __syncthreads();
    T acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv == 0) {
        T *const junk = junkbuf + lane;
        for (int rep = 0; rep < reps; rep++) {
            T a[4], lprev = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) a[c] = tile[(4 * q + c) * 16 + i] + (T)rep * 1e-9;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int kq = k >> 2, kc = k & 3;
                colx4[q][i] = a[kc];
                asm volatile("" ::: "memory");
                const T lraw = colx4[kq][i];
                T y[4];
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = colx4[kq][4 * q + c];
                asm volatile("" ::: "memory");
                if (k > 0) {
                    *((q == ((k - 1) >> 2)) ? &Ad[k - 1][i] : junk) = lprev;
                    __hip_atomic_store(&prog[lane], 16 * rep + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                asm volatile("" ::: "memory");
                const T r = ba_rcp(ba_readlane(a[kc], 16 * kq + k));
                __builtin_amdgcn_sched_barrier(0);
                const T lm = (i > k) ? lraw : (T)0;
                const T l = lm * r;
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] -= l * y[c];
                lprev = l;
            }
            __hip_atomic_store(&prog[lane], 16 * rep + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int c = 0; c < 4; c++) acc += a[c];
        }
        if (lane == 0) __hip_atomic_store(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (wv == 1) {
        if (MODE & 1) {
            int pg = 0;
            for (int rep = 0; rep < reps; rep++) {
                T w[4];
#pragma unroll
                for (int c = 0; c < 4; c++) w[c] = (4 * q + c == i) ? (T)1 : (T)0;
#pragma unroll
                for (int k = 0; k < 15; k++) {
                    T wk[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) wk[c] = dppb_k(w[c], k);
                    T lr = __hip_atomic_load(&Ad[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    while (pg < 16 * rep + k + 1) {
                        pg = __hip_atomic_load(&prog[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        lr = __hip_atomic_load(&Ad[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    const T l = (i > k) ? lr * 1e-3 : (T)0;
#pragma unroll
                    for (int c = 0; c < 4; c++) w[c] -= l * wk[c];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) acc += w[c];
            }
        }
    } else {
        typedef ba_acc<T>::type acc_t;
        acc_t c4;
        for (int v = 0; v < 4; v++) c4[v] = 0;
        const int li = lane & 15, lk = lane >> 4, h = wv - 2;
        if (MODE & 2) {
            T x = 1e-3 * lane, y2 = 1e-4 * lane;
            while (!__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
                for (int m = 0; m < 16; m++) c4 = ba_mfma(x, y2, c4);
            }
        }
        if (MODE & 4) {
            while (!__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                T la[4], yb[4];
#pragma unroll
                for (int v = 0; v < 4; v++) c4[v] = Big[32 * h + ba_crow<T>(lk, v)][16 + li];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) { la[kk] = Big[32 * h + 4 * kk + lk][32 + li]; yb[kk] = Big[32 * h + 16 + 4 * kk + lk][li]; }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) c4 = ba_mfma(la[kk], yb[kk], c4);
#pragma unroll
                for (int v = 0; v < 4; v++) Big[32 * h + ba_crow<T>(lk, v)][16 + li] = c4[v] * 1e-3;
            }
        }
        for (int v = 0; v < 4; v++) acc += c4[v];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[tid] = acc;
    if (lane == 0) out[wv] = (long long)(t1 - t0);
}
int main()
{
    std::vector<double> h(256);
    for (int c = 0; c < 16; c++) for (int r = 0; r < 16; r++) h[c * 16 + r] = (r == c ? 20.0 : 0.0) + 1.0 / (1 + r + c);
    double *tile, *sink; long long *d, t;
    hipMalloc(&tile, 2048); hipMalloc(&sink, 4096); hipMalloc(&d, 64);
    hipMemcpy(tile, h.data(), 2048, hipMemcpyHostToDevice);
    const int reps = 2000;
    const char *names[] = {"V0 wave-0 loop as shipped", "V1 no L/progress stores", "V2 V1 with ONE Newton step", "V3 V1 without the LDS exchange", "V4 V1 without readlane/rcp", "V5 DPP row + bpermute column", "V6 V5 without L/progress stores", "V7 V5 hand-ordered, stores deferred", "V8 V7 without L/progress stores", "V9 row swaps + DPP, stores deferred", "V10 V9 without L/progress stores", "V11 V0, compiler-only ordering", "V12 V11 without L/progress stores", "V13 V11, stores behind next loads", "V14 V13, Newton before the load wait", "V15 V14 + inverse in the same wave"};
#define RUN(V, TH) hipLaunchKernelGGL(probe<V>, dim3(1), dim3(TH), 0, 0, d, sink, tile, reps); hipDeviceSynchronize(); hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost); printf("%-36s (%3d threads): %.1f cycles per pivot\n", names[V], TH, t / (double)reps / 16);
    RUN(0, 64) RUN(0, 256) RUN(1, 64) RUN(2, 64) RUN(3, 64) RUN(4, 64) RUN(5, 64) RUN(6, 64) RUN(7, 64) RUN(8, 64) RUN(9, 64) RUN(10, 64) RUN(11, 64) RUN(12, 64) RUN(13, 64) RUN(14, 64) RUN(15, 64)
    long long t4[4];
#define RUN2(W, M, label) hipLaunchKernelGGL((probe2<W, M>), dim3(1), dim3(256), 0, 0, d, sink, tile, reps); hipDeviceSynchronize(); hipMemcpy(t4, d, 32, hipMemcpyDeviceToHost); printf("%-52s: wave 0 %.1f, wave %d %.1f cycles per pivot\n", label, t4[0] / (double)reps / 16, W, t4[W] / (double)reps / 16);
    RUN2(1, 1, "look-ahead factor wave alone (no L stores)")
    RUN2(1, 2, "inverse wave alone (no polling)")
    RUN2(1, 0, "both, inverse on wave 1")
    RUN2(2, 0, "both, inverse on wave 2")
    RUN2(3, 0, "both, inverse on wave 3")
#define RUN3(M, label) hipLaunchKernelGGL((probe3<M>), dim3(1), dim3(256), 0, 0, d, sink, tile, reps); hipDeviceSynchronize(); hipMemcpy(t4, d, 32, hipMemcpyDeviceToHost); printf("%-60s: factor wave %.1f cycles per pivot\n", label, t4[0] / (double)reps / 16);
    RUN3(0, "shipped loop, other waves idle")
    RUN3(1, "... + inverse wave following through the progress word")
    RUN3(2, "... + waves 2, 3 issuing f64 MFMAs")
    RUN3(4, "... + waves 2, 3 doing LDS tile updates")
    RUN3(3, "... + inverse wave + MFMAs")
    RUN3(5, "... + inverse wave + LDS tile updates")
    return 0;
}
