// Dev tool: times the dense Householder QR of ba_qr.hip.h on a random tall matrix (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form [-DBA_QR_STAMP] -I bundleadjustment_benchmarks_amd/csrc scripts/bench_qr.hip -o scripts/bench_qr.bin
// usage: bench_qr.bin [m = 181633] [n = 351]   (config 3's J2bot; fp32)
#include "ba_mfma.hip.h"
#define BA_REC 32 /* (ba_kernels.hip.h: scalars per observation record; only k_qrkit_build, unused here, needs it) */
#ifdef BA_QR_STAMP2
__device__ long long ba_qr_fine[16];
#endif
#ifdef BA_QR_STAMP
__device__ long long ba_qr_stamp[40];
__device__ long long ba_qr_busy[8 * 32]; // [step][wave]: cycles from the barrier exit to the end of the wave's work of that step // wave 0 of the one-workgroup launch of a panel: time at every step's barrier exit, then the end
#endif
#include "ba_qr.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float T;
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 181633, n = argc > 2 ? atoi(argv[2]) : 351;
    const size_t lda = (size_t)m + 64;
    std::vector<T> h(lda * (n + 1), 0);
    srand(3);
    for (int c = 0; c <= n; c++)
        for (int r = 0; r < m; r++) h[(size_t)c * lda + r] = (T)(rand() / (double)RAND_MAX - 0.5) * ((r % 37 == c % 37) ? 8 : 1);
    T *A, *A0, *tau, *y;
    const size_t tstride = (size_t)((m + ba_qr_cfg<T>::CH - 1) / ba_qr_cfg<T>::CH + 2) * BA_QR_PB * BA_QR_PB;
    CK(hipMalloc(&A, sizeof(T) * h.size())); CK(hipMalloc(&A0, sizeof(T) * h.size())); CK(hipMalloc(&tau, sizeof(T) * 2 * BA_QR_TAU_LEVELS * tstride)); CK(hipMalloc(&y, sizeof(T) * (n + 64)));
    CK(hipMemcpy(A0, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    hipStream_t st, st2, st3; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st3, hipStreamNonBlocking));
    hipEvent_t e0, e1, ea, eb, ec; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ec, hipEventDisableTiming));
    ba_qr_side sd; // BENCH_QR_STREAMS = 1 | 2 | 3 (default): one stream, trailing updates beside the chain, look-ahead on a third one
    { const int ns = getenv("BENCH_QR_STREAMS") ? atoi(getenv("BENCH_QR_STREAMS")) : 3;
      if (ns >= 2) { sd.st2 = st2; sd.ev_chunk = ea; sd.ev_apply = eb; }
      if (ns >= 3) { sd.st3 = st3; sd.ev_next = ec; }
      printf("streams: %d\n", ns); }
    float tot = 0;
    const int reps = 5;
    for (int rep = 0; rep <= reps; rep++) {
        CK(hipMemcpyAsync(A, A0, sizeof(T) * h.size(), hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e0, st));
        ba_qr_solve<T>(st, A, lda, m, n, tau, tstride, y, sd);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) tot += ms;
    }
    std::vector<T> yh(n);
    CK(hipMemcpy(yh.data(), y, sizeof(T) * n, hipMemcpyDeviceToHost));
    // normal-equations residual |A^T (A y - b)| / |A^T b| in double
    std::vector<double> r(m), g(n, 0.0), g0(n, 0.0);
    for (int i = 0; i < m; i++) { double a = -(double)h[(size_t)n * lda + i]; for (int c = 0; c < n; c++) a += (double)h[(size_t)c * lda + i] * yh[c]; r[i] = a; }
    double gn = 0, g0n = 0;
    for (int c = 0; c < n; c++) { double a = 0, b = 0; for (int i = 0; i < m; i++) { a += (double)h[(size_t)c * lda + i] * r[i]; b += (double)h[(size_t)c * lda + i] * h[(size_t)n * lda + i]; } gn += a * a; g0n += b * b; }
    printf("QR %d x %d fp32: %.3f ms per factorisation + solve = %.1f TFLOP/s; |A'(Ay-b)| / |A'b| = %.2e\n", m, n, tot / reps, 2.0 * m * n * n / (tot / reps * 1e-3) / 1e12,
           std::sqrt(gn / g0n));
#ifdef BA_QR_STAMP2
    { long long hf[16]; CK(hipMemcpyFromSymbol(hf, HIP_SYMBOL(ba_qr_fine), sizeof(hf)));
      printf("last one-workgroup chunk launch, step 8.  owner (wave 0): sum of squares %lld, wave reduction %lld, scalars (sqrt, reciprocals) %lld, v to LDS %lld, column to memory %lld cycles\n",
             hf[1] - hf[0], hf[2] - hf[1], hf[3] - hf[2], hf[4] - hf[3], hf[5] - hf[4]);
      printf("wave 5 (three live columns): v from LDS %lld, dot products %lld, three wave reductions %lld, updates %lld cycles\n", hf[7] - hf[6], hf[8] - hf[7], hf[9] - hf[8], hf[10] - hf[9]); }
#endif
#ifdef BA_QR_STAMP
    long long hs[40]; CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(ba_qr_stamp), sizeof(hs)));
    printf("last one-workgroup chunk launch, wave 0: cycles from kernel start to the exit of the barrier of step j, j = 0 .. 31, then loop end, then kernel end:\n");
    for (int j = 0; j < 35; j++) printf("%lld ", hs[j] - hs[36]);
    { long long hb[256]; CK(hipMemcpyFromSymbol(hb, HIP_SYMBOL(ba_qr_busy), sizeof(hb)));
      printf("\nbusy cycles per step and wave (the owner of step j + 1 is wave (j + 1) %% 8):\n");
      for (int j = 0; j < 31; j++) { printf(" j=%2d:", j); for (int w = 0; w < 8; w++) printf(" %5lld", hb[8 * j + w]); printf("\n"); } }
    printf("per step:");
    for (int j = 1; j < 32; j++) printf(" %lld", hs[j] - hs[j - 1]);
    printf("\n");
#endif
    return 0;
}
