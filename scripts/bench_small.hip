// Dev tool: the one-workgroup LDL^T + solve of ba_small.hip.h on random SPD matrices: residual and time per launch
// (not part of the product: the kernel was measured and NOT adopted, profiles/EXPERIMENTS.md 1.5).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I bundleadjustment_benchmarks_amd/csrc -I scripts scripts/bench_small.hip -o scripts/bench_small.bin
#include "experiments/ldlt_small.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <typename T> static int run(int D)
{
    const int Dp = ((D + 2 + 63) / 64) * 64, ld = Dp + 64;
    std::vector<double> h((size_t)ld * Dp, 0.0);
    srand(D);
    std::vector<double> v((size_t)D * 8);
    for (auto &x : v) x = rand() / (double)RAND_MAX - 0.5;
    for (int c = 0; c < D; c++)
        for (int r = c; r < D; r++) {
            double a = (r == c) ? 4.0 : 0.0;
            for (int k = 0; k < 8; k++) a += v[(size_t)r * 8 + k] * v[(size_t)c * 8 + k];
            h[(size_t)c * ld + r] = a;
        }
    for (int c = 0; c < D; c++) h[(size_t)c * ld + D] = rand() / (double)RAND_MAX; // rhs row
    for (int c = D; c < Dp; c++) h[(size_t)c * ld + c] = 1.0;
    std::vector<T> ht(h.size());
    for (size_t i = 0; i < h.size(); i++) ht[i] = (T)h[i];
    T *S, *x;
    CK(hipMalloc(&S, sizeof(T) * ht.size())); CK(hipMalloc(&x, sizeof(T) * Dp));
    CK(hipMemcpy(S, ht.data(), sizeof(T) * ht.size(), hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, sizeof(T) * Dp));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    hipLaunchKernelGGL((k_ldlt_small<T>), dim3(1), dim3(512), 0, st, D, ld, (const T *)S, x);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_ldlt_small<T>), dim3(1), dim3(512), 0, st, D, ld, (const T *)S, x);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<T> xs(D);
    CK(hipMemcpy(xs.data(), x, sizeof(T) * D, hipMemcpyDeviceToHost));
    double rn = 0, bn = 0;
    for (int r = 0; r < D; r++) {
        double a = 0;
        for (int c = 0; c < D; c++) a += (c <= r ? h[(size_t)c * ld + r] : h[(size_t)r * ld + c]) * (double)xs[c];
        const double b = h[(size_t)r * ld + D];
        rn += (a - b) * (a - b); bn += b * b;
    }
    const double res = std::sqrt(rn / bn);
#ifdef BA_SMALL_STAMP
    { long long hs[2][16][6]; CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(ba_small_stamp), sizeof(hs)));
      const int nct = (D + 15) / 16;
      for (int s = 0; s < nct; s++)
          printf("  step %2d  wave0: wait %5lld pivots %5lld wait %5lld tail %5lld | wave1: wait %5lld Y+sync %5lld first %5lld rest %5lld | step: wave0 %5lld wave1 %5lld\n", s,
                 hs[0][s][1] - hs[0][s][0], hs[0][s][2] - hs[0][s][1], hs[0][s][3] - hs[0][s][2], hs[0][s][4] - hs[0][s][3],
                 hs[1][s][1] - hs[1][s][0], hs[1][s][2] - hs[1][s][1], hs[1][s][3] - hs[1][s][2], hs[1][s][4] - hs[1][s][3],
                 hs[0][s][4] - hs[0][s][0], hs[1][s][4] - hs[1][s][0]);
      for (int w = 0; w < 2; w++)
          printf("  wave %d: load %lld  factor %lld  (wait at the end %lld)  sweep %lld  total %lld ticks\n", w, hs[w][15][1] - hs[w][15][0], hs[w][15][2] - hs[w][15][1],
                 hs[w][15][3] - hs[w][15][2], hs[w][15][4] - hs[w][15][3], hs[w][15][4] - hs[w][15][0]); }
#endif
    printf("%s D=%3d: %.2f us per launch, residual %.2e%s\n", sizeof(T) == 8 ? "f64" : "f32", D, ms / reps * 1e3, res,
           res < (sizeof(T) == 8 ? 1e-13 : 1e-4) ? "" : "  <-- WRONG");
    CK(hipFree(S)); CK(hipFree(x));
    return res < (sizeof(T) == 8 ? 1e-13 : 1e-4) ? 0 : 1;
}
int main(int argc, char **argv)
{
    int bad = 0;
    if (argc > 1) { for (int i = 1; i < argc; i++) { bad += run<double>(atoi(argv[i])); bad += run<float>(atoi(argv[i])); } return bad; }
    const int Ds[] = {9, 15, 16, 17, 18, 27, 31, 32, 33, 45, 63, 64, 65, 99, 117, 128, 135, 144, 160, 176, 189, 192, 200, 207};
    for (int D : Ds) { bad += run<double>(D); bad += run<float>(D); }
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
