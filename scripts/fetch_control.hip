// Dev tool (evidence, not product): what rocprofv3's FETCH_SIZE reports on gfx950 for the two access patterns of this repo, against
// byte counts known by construction.  MI355X_MICROARCH.md says the counter tallies 128-byte fabric requests as 64 bytes for wide
// coalesced streaming reads (x2 correction); whether that also holds for k_schur_pairs' 8-byte raw-buffer gathers of 256-byte
// records decides which of its two traffic figures (raw / x2) is the right one.
//   k_stream16  : every lane reads 16 bytes, consecutive lanes consecutive addresses, NB bytes in all (>> Infinity Cache), once
//   k_gather8   : k_schur_pairs' pattern -- lane (i < 9, k) reads three doubles at byte 24 i of record perm[4 t + k] (256-byte records,
//                 random permutation, every record once): 216 of the 256 bytes of a record, all four of its 64-byte sectors
// hipcc --offload-arch=gfx950 -O3 scripts/fetch_control.hip -o scripts/fetch_control.bin
// rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o fc --output-format csv -- scripts/fetch_control.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const float4 *__restrict__ p, size_t n16, float *__restrict__ out)
{
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_gather8(const double *__restrict__ rec, unsigned rec_bytes, const int *__restrict__ perm, int nrec, double *__restrict__ out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(rec), 0, (int)rec_bytes, 0x00020000);
    const int lane = threadIdx.x & 63, i = lane & 15, k = lane >> 4;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    double acc = 0;
    for (int t = wave; 4 * t < nrec; t += nwaves) { // four records per wave and trip, lane (i, k) on record 4 t + k
        const int e = 4 * t + k;
        const unsigned off = (e < nrec && i < 9) ? (unsigned)perm[e] * 256u + 24u * i : 0xfffffff0u - 64 * 8;
#pragma unroll
        for (int m = 0; m < 3; m++) acc += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(off + 8 * m), 0, 0));
    }
    if (acc == 12345.678) out[0] = acc;
}

int main()
{
    const size_t NB = 1ull << 30; // 1 GiB: four times the Infinity Cache
    float *buf; double *out; int *perm;
    CK(hipMalloc(&buf, NB)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, NB));
    const int nrec = (int)(NB / 256);
    std::vector<int> h(nrec);
    std::iota(h.begin(), h.end(), 0);
    std::shuffle(h.begin(), h.end(), std::mt19937(7));
    CK(hipMalloc(&perm, sizeof(int) * nrec));
    CK(hipMemcpy(perm, h.data(), sizeof(int) * nrec, hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stream16, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)buf, NB / 16, (float *)out);
        CK(hipEventRecord(e1));
        // (the gather only reaches the first 4 GiB - 1 of a raw buffer: 1 GiB is fine)
        hipLaunchKernelGGL(k_gather8, dim3(256 * 8), dim3(256), 0, 0, (const double *)buf, (unsigned)(NB - 1), perm, nrec, out);
        CK(hipEventRecord(e2));
        CK(hipDeviceSynchronize());
        float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        printf("rep %d: k_stream16 %zu bytes in %.3f ms = %.2f TB/s | k_gather8 %d records x 256 B = %zu bytes (216 B of each requested) in %.3f ms = %.2f TB/s\n", rep, NB,
               a, NB / (a * 1e-3) / 1e12, nrec, (size_t)nrec * 256, b, (size_t)nrec * 256 / (b * 1e-3) / 1e12);
    }
    return 0;
}
