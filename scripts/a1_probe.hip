// Dev tool: what bounds the 16x16 diagonal-tile pivot loop (A1 of ba_panel_body) on gfx950?  Single wave, variants.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bundleadjustment_benchmarks_amd/csrc scripts/a1_probe.hip -o scripts/a1_probe.bin
#include "ba_dense.hip.h"
#include <cstdio>
#include <vector>
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
        T a[4];
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = tile[(4 * q + c) * 16 + i] + (T)rep * 1e-9;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int kq = k >> 2, kc = k & 3;
            T lraw, y[4], r;
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
        for (int c = 0; c < 4; c++) acc += a[c];
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
int main()
{
    std::vector<double> h(256);
    for (int c = 0; c < 16; c++) for (int r = 0; r < 16; r++) h[c * 16 + r] = (r == c ? 20.0 : 0.0) + 1.0 / (1 + r + c);
    double *tile, *sink; long long *d, t;
    hipMalloc(&tile, 2048); hipMalloc(&sink, 4096); hipMalloc(&d, 64);
    hipMemcpy(tile, h.data(), 2048, hipMemcpyHostToDevice);
    const int reps = 2000;
    const char *names[] = {"V0 wave-0 loop as shipped", "V1 no L/progress stores", "V2 V1 with ONE Newton step", "V3 V1 without the LDS exchange", "V4 V1 without readlane/rcp"};
#define RUN(V, TH) hipLaunchKernelGGL(probe<V>, dim3(1), dim3(TH), 0, 0, d, sink, tile, reps); hipDeviceSynchronize(); hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost); printf("%-36s (%3d threads): %.1f cycles per pivot\n", names[V], TH, t / (double)reps / 16);
    RUN(0, 64) RUN(0, 256) RUN(1, 64) RUN(2, 64) RUN(3, 64) RUN(4, 64)
    long long t4[4];
#define RUN2(W, M, label) hipLaunchKernelGGL((probe2<W, M>), dim3(1), dim3(256), 0, 0, d, sink, tile, reps); hipDeviceSynchronize(); hipMemcpy(t4, d, 32, hipMemcpyDeviceToHost); printf("%-52s: wave 0 %.1f, wave %d %.1f cycles per pivot\n", label, t4[0] / (double)reps / 16, W, t4[W] / (double)reps / 16);
    RUN2(1, 1, "look-ahead factor wave alone (no L stores)")
    RUN2(1, 2, "inverse wave alone (no polling)")
    RUN2(1, 0, "both, inverse on wave 1")
    RUN2(2, 0, "both, inverse on wave 2")
    RUN2(3, 0, "both, inverse on wave 3")
    return 0;
}
