// Dev tool: latency of the pivot recurrence pieces on gfx950 (single wave).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rcp2(double d) { double r = __builtin_amdgcn_rcp(d); r = fma(fma(-d, r, 1.0), r, r); r = fma(fma(-d, r, 1.0), r, r); return r; }
template <int MODE> __global__ void probe(long long *out, double *sink, int iters)
{
    __shared__ double lds[128];
    double x = 1.0 + threadIdx.x * 1e-6, acc = 0;
    lds[threadIdx.x & 127] = x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) x = __builtin_amdgcn_rcp(x) + 1.0;                 // rcp + add
        if (MODE == 1) x = rcp2(x) + 1.0;                                // rcp + 2 newton + add
        if (MODE == 2) x = 1.0 / x + 1.0;                                // full division
        if (MODE == 3) { lds[threadIdx.x & 63] = x; x = lds[(threadIdx.x + 1) & 63] + 1e-9; } // LDS write -> read (other lane) chain
        if (MODE == 4) { int lo = __builtin_amdgcn_readlane(__double2loint(x), it & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), it & 63); x = __hiloint2double(hi, lo) * 1.0000001 + 1e-9; }
        if (MODE == 5) { x = fma(x, 1.0000001, 1e-9); __syncthreads(); }
        if (MODE == 6) { x = __shfl(x, (threadIdx.x + 1) & 63) * 1.0000001; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[threadIdx.x] = x + acc;
    if (threadIdx.x == 0) out[0] = (long long)(t1 - t0);
}
int main()
{
    long long *d, h; double *sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 8 * 1024);
    const char *names[] = {"v_rcp_f64 + add", "rcp + 2 Newton + add", "full f64 division + add", "LDS write -> read(other lane) + add", "readlane(64-bit) + fma", "fma + __syncthreads (4 waves)", "__shfl (bpermute) + mul"};
    const int it = 20000;
#define RUN(M, TH) hipLaunchKernelGGL(probe<M>, dim3(1), dim3(TH), 0, 0, d, sink, it); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf("%-40s: %.1f cycles per iteration\n", names[M], h / (double)it);
    RUN(0, 64) RUN(1, 64) RUN(2, 64) RUN(3, 64) RUN(4, 64) RUN(5, 256) RUN(6, 64)
    return 0;
}
