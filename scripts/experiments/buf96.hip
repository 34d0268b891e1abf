// Experiment (VERDICT r3 item 8 ii): does a 12-byte raw-buffer load (buffer_load_dwordx3) return the same three floats as three 4-byte
// loads, at every 4-byte-aligned offset of a record, with the descriptor word 0x00020000 the Schur kernel uses and with out-of-range
// offsets as masks?   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/experiments/buf96.hip -o scripts/experiments/buf96.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v3u __attribute__((ext_vector_type(3)));
__global__ void k(const float *rec, unsigned bytes, const unsigned *offs, float *out3, float *out1, int n, int word3)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rec), 0, (int)bytes, word3);
    const unsigned off = offs[i];
#ifdef VIA_OTHER_VECTOR_TYPE
    // round 3's form: the builtin's result assigned to a 3-vector type of one's own.  It compiles -- to ONE buffer_load_dword whose value
    // is copied into all three elements (ROCm 7.2 clang: a conversion, not a bit cast; see the ISA) -- hence "wrong sums".
    const v3u v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, 0, 0);
    out3[3 * i] = __builtin_bit_cast(float, v.x); out3[3 * i + 1] = __builtin_bit_cast(float, v.y); out3[3 * i + 2] = __builtin_bit_cast(float, v.z);
#else
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, 0, 0); // the builtin's own vector type: buffer_load_dwordx3
    out3[3 * i] = __builtin_bit_cast(float, v[0]); out3[3 * i + 1] = __builtin_bit_cast(float, v[1]); out3[3 * i + 2] = __builtin_bit_cast(float, v[2]);
#endif
    for (int m = 0; m < 3; m++) out1[3 * i + m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(off + 4 * m), 0, 0));
}
int main()
{
    const int nrec = 4096, n = 1 << 16;
    std::vector<float> h(nrec * 32);
    for (size_t q = 0; q < h.size(); q++) h[q] = 1.0f + (float)q;
    std::vector<unsigned> offs(n);
    unsigned s = 12345;
    const unsigned bytes = nrec * 32 * 4, OOB = 0xfffffff0u - 64 * 4;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        const unsigned rec = (s >> 8) % nrec, lane = (s >> 24) % 11; // 3 * lane floats into the record, like lane_off = 3 i SZ
        offs[i] = (i % 7 == 3) ? OOB : (i % 97 == 5) ? bytes - 8 /* straddles the end */ : rec * 128 + 12 * lane;
    }
    float *d, *o3, *o1; unsigned *dof;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o3, n * 12); hipMalloc(&o1, n * 12); hipMalloc(&dof, n * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dof, offs.data(), n * 4, hipMemcpyHostToDevice);
  for (int word3 : {0x00020000, 0x00027000, 0x00020FAC, 0x00027FAC, 0x00068000, 0x0006FFAC}) { // DATA_FORMAT 32 (the Schur kernel's) .. 32_32_32, with / without DST_SEL, NUM_FORMAT
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, bytes, dof, o3, o1, n, word3);
    printf("descriptor word 3 = 0x%08x: ", word3);
    std::vector<float> a(3 * n), b(3 * n);
    hipMemcpy(a.data(), o3, n * 12, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, n * 12, hipMemcpyDeviceToHost);
    int bad = 0, bad_in = 0, bad_oob = 0, bad_end = 0;
    for (int i = 0; i < n; i++)
        for (int m = 0; m < 3; m++)
            if (a[3 * i + m] != b[3 * i + m]) {
                bad++;
                if (offs[i] == OOB) bad_oob++; else if (offs[i] == bytes - 8) bad_end++; else bad_in++;
                if (bad <= 2) printf("i %d off %u (rec %u + %u) m %d: b96 %g, b32 %g\n", i, offs[i], offs[i] / 128, offs[i] % 128, m, a[3 * i + m], b[3 * i + m]);
            }
    printf("mismatches: %d of %d (in range %d, masked offsets %d, straddling the end %d)\n", bad, 3 * n, bad_in, bad_oob, bad_end);
  }
    return 0;
}
