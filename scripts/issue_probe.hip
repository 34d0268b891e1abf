// Dev tool: issue cost (cycles per wave-instruction, one wave per SIMD) of independent f64 / f32 ops on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void probe(long long *out, double *sink, int iters)
{
    __shared__ double lds[512];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.0;
    __syncthreads();
    double a[8]; float f[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
    const double b = 1.000001; const float bf = 1.000001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = fma(a[i], b, 1e-9); }
        if (MODE == 1) { _Pragma("unroll") for (int i = 0; i < 8; i++) f[i] = fmaf(f[i], bf, 1e-9f); }
        if (MODE == 2) { _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = (a[i] > (double)it) ? a[i] : b; }
        if (MODE == 3) { _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] += lds[(threadIdx.x + 8 * it + i) & 511]; }
        if (MODE == 4) { _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = a[i] * b; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i] + f[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = (long long)(t1 - t0);
}
int main()
{
    long long *d, h; double *sink;
    hipMalloc(&d, 8 * 64); hipMalloc(&sink, 8 * 4096);
    const char *names[] = {"f64 fma (8 independent)", "f32 fma (8 independent)", "f64 select (cmp + 2 cndmask)", "lds read b64 + f64 add", "f64 mul"};
    const int it = 20000;
    for (int threads : {256, 512, 1024}) {
#define RUN(M) hipLaunchKernelGGL(probe<M>, dim3(1), dim3(threads), 0, 0, d, sink, it); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); \
    printf("%4d threads (%d waves/SIMD) %-32s: %.2f cycles per wave-op-group of 8 -> %.2f per op\n", threads, threads / 256, names[M], h / (double)it, h / (double)it / 8);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    }
    return 0;
}
