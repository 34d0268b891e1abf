// Dev tool: in-kernel shader clock (s_memtime ticks per 100 MHz s_memrealtime tick), light load vs heavy load.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) a = fma(a, b, 1e-9);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = (long long)(t1 - t0); out[2 * blockIdx.x + 1] = (long long)(r1 - r0); }
    if (a == 12345.678) out[0] = 0;
}
int main()
{
    long long *d, h[2];
    hipMalloc(&d, 16 * 4096);
    for (int grid : {1, 1, 36, 2048, 1}) {
        for (int it : {2000, 200000}) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, it);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("grid %5d iters %7d: %lld shader ticks / %lld realtime ticks(100MHz) -> %.3f GHz ; %.1f cycles per dependent f64 fma\n", grid, it, h[0], h[1],
                   h[0] / (double)h[1] * 0.1, h[0] / (double)it);
        }
    }
    return 0;
}
