// ba_mfma.hip.h -- the 16x16x4 matrix-core instruction of gfx950 for Scalar = double / float, shared by the dense
// factorisation (ba_dense.hip.h) and the Schur-complement assembly (ba_kernels.hip.h).
//
// Fragment maps (cdna_hip_programming.md s3): A[i][k]: lane l holds i = l & 15, k = l >> 4; B[k][j]: lane l holds
// k = l >> 4, j = l & 15; C/D: col = l & 15, row = (l >> 4) + 4 v for f64 and 4 (l >> 4) + v for f32 (v = 0..3).
#ifndef BA_MFMA_HIP_H
#define BA_MFMA_HIP_H

#include <hip/hip_runtime.h>

// Device error word (one scalar slot of the solver): written by a kernel whose bounded in-launch hand-off wait ran out.
#define BA_DEVERR_ROW_FLAG 1 /* k_ldlt_step: a panel workgroup never saw its rows' look-ahead update announced */
#define BA_DEVERR_SWEEP 1024 /* k_ldlt_backflow: an unknown of a later group was never published (the error words of the shards are SUMMED by the
                                scalar all-reduce: row-flag time-outs count below 1024, this one above) */

#define BA_DEVERR_DIVERGED 3 /* k_lm_control: the shards of a sharded solve did not take the same accept / stop decision */

typedef double ba_d4 __attribute__((ext_vector_type(4)));
typedef float ba_f4 __attribute__((ext_vector_type(4)));

template <typename T> struct ba_acc;
template <> struct ba_acc<double> { typedef ba_d4 type; };
template <> struct ba_acc<float> { typedef ba_f4 type; };

__device__ __forceinline__ ba_d4 ba_mfma(double a, double b, ba_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ba_f4 ba_mfma(float a, float b, ba_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename T> __device__ __forceinline__ int ba_crow(int lk, int v) { return sizeof(T) == 8 ? lk + 4 * v : 4 * lk + v; }


// LDS hand-off between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order, but the
// compiler must be told that the load below reads what OTHER lanes stored above (it otherwise hoists the load over the
// lane-predicated store, which is legal for a single thread).
__device__ __forceinline__ void ba_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Ordering for the compiler only.  The LDS executes the instructions of ONE wave in order, so a load behind a store of the
// same wave sees it without any s_waitcnt in between (a wavefront fence would emit one and put the store's round trip in
// front of the load's: ~45 cycles per pivot in the loops below).
__device__ __forceinline__ void ba_wave_lds_order() { asm volatile("" ::: "memory"); }

// ---- wave reduction by DPP (dense back sweep, QRKIT QR) -------------------------------------------------------------
__device__ __forceinline__ float ba_readlane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ double ba_readlane63(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}


// v of lane `k` (wave-uniform k) in every lane
__device__ __forceinline__ float ba_readlane_dyn(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); }
__device__ __forceinline__ double ba_readlane_dyn(double v, int k)
{
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), k), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), k);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}

// Sum over the 64 lanes, the same value returned to every lane: four butterfly steps inside a row of 16 lanes by DPP (quad
// permutes, half-row and row mirrors), the two row broadcasts that carry the row sums to lane 63, one v_readlane -- seven vector
// instructions, no LDS crossbar (a __shfl_xor tree is six ds_bpermute round trips, one after the other).
template <int CTRL, int ROW_MASK, typename T> __device__ __forceinline__ T ba_dpp_add(T v)
{
    return v + __builtin_amdgcn_update_dpp((T)0, v, CTRL, ROW_MASK, 0xf, false);
}
template <typename T> __device__ __forceinline__ T ba_wave_sum_all(T v)
{
    v = ba_dpp_add<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
    v = ba_dpp_add<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
    v = ba_dpp_add<0x141, 0xf>(v); // row_half_mirror
    v = ba_dpp_add<0x140, 0xf>(v); // row_mirror: every lane holds the sum of its row of 16
    v = ba_dpp_add<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v = ba_dpp_add<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return ba_readlane63(v);
}

// FOUR sums over the 64 lanes at once (k_qr_chunk: the dot products of a reflector with a wave's four live columns): the first two
// butterfly steps hand each lane ONE of the four values (lane & 3 picks it: a select pair and one quad-permute add per two values),
// then the remaining steps run once instead of four times -- two rotations inside the row of 16 (row_ror 4, 8: they keep lane & 3),
// v_permlane16_swap and v_permlane32_swap across the rows (gfx950) -- and lanes 0 .. 3 hold the four totals: four v_readlane.
// Eleven cross-lane instructions instead of twenty-eight; the dependent chain is six steps either way.
__device__ __forceinline__ float ba_swap_add(float v, bool half32)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = half32 ? __builtin_amdgcn_permlane32_swap(u, u, false, false) : __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ double ba_swap_add(double v, bool half32)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto rl = half32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = half32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double a0 = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[0] << 32) | (unsigned)rl[0]);
    const double a1 = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[1] << 32) | (unsigned)rl[1]);
    return a0 + a1;
}
template <typename T> __device__ __forceinline__ void ba_wave_sum4_all(T (&x)[4], int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2;
    const T k01 = o1 ? x[1] : x[0], s01 = o1 ? x[0] : x[1];
    const T k23 = o1 ? x[3] : x[2], s23 = o1 ? x[2] : x[3];
    const T r01 = k01 + __builtin_amdgcn_update_dpp((T)0, s01, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
    const T r23 = k23 + __builtin_amdgcn_update_dpp((T)0, s23, 0xB1, 0xf, 0xf, false);
    const T kb = o2 ? r23 : r01, sb = o2 ? r01 : r23;
    T r = kb + __builtin_amdgcn_update_dpp((T)0, sb, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]: lane l holds value l & 3 of its quad
    r = ba_dpp_add<0x124, 0xf>(r);                                               // row_ror:4
    r = ba_dpp_add<0x128, 0xf>(r);                                               // row_ror:8: ... of its row of 16
    r = ba_swap_add(r, false);                                                   // rows 0 + 1, 2 + 3
    r = ba_swap_add(r, true);                                                    // both halves
#pragma unroll
    for (int c = 0; c < 4; c++) x[c] = ba_readlane_dyn(r, c);
}

#endif
