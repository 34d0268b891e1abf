#include <hip/hip_runtime.h>
__global__ void k1(const float *rec, float *o) // elements bit-cast to float one by one
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rec), 0, 1 << 20, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)(threadIdx.x * 12), 0, 0);
    o[3 * threadIdx.x] = __builtin_bit_cast(float, v[0]); o[3 * threadIdx.x + 1] = __builtin_bit_cast(float, v[1]); o[3 * threadIdx.x + 2] = __builtin_bit_cast(float, v[2]);
}
__global__ void k2(const float *rec, unsigned *o) // elements stored as they are
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rec), 0, 1 << 20, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)(threadIdx.x * 12), 0, 0);
    o[3 * threadIdx.x] = v[0]; o[3 * threadIdx.x + 1] = v[1]; o[3 * threadIdx.x + 2] = v[2];
}
__global__ void k3(const float *rec, float *o, int n) // inside a bounds check like the experiment
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rec), 0, 1 << 20, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r, i * 12, 0, 0);
    o[3 * i] = __builtin_bit_cast(float, v[0]); o[3 * i + 1] = __builtin_bit_cast(float, v[1]); o[3 * i + 2] = __builtin_bit_cast(float, v[2]);
}
