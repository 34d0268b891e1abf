// ba_mfma.hip.h -- the 16x16x4 matrix-core instruction of gfx950 for Scalar = double / float, shared by the dense
// factorisation (ba_dense.hip.h) and the Schur-complement assembly (ba_kernels.hip.h).
//
// Fragment maps (cdna_hip_programming.md s3): A[i][k]: lane l holds i = l & 15, k = l >> 4; B[k][j]: lane l holds
// k = l >> 4, j = l & 15; C/D: col = l & 15, row = (l >> 4) + 4 v for f64 and 4 (l >> 4) + v for f32 (v = 0..3).
#ifndef BA_MFMA_HIP_H
#define BA_MFMA_HIP_H

#include <hip/hip_runtime.h>

typedef double ba_d4 __attribute__((ext_vector_type(4)));
typedef float ba_f4 __attribute__((ext_vector_type(4)));

template <typename T> struct ba_acc;
template <> struct ba_acc<double> { typedef ba_d4 type; };
template <> struct ba_acc<float> { typedef ba_f4 type; };

__device__ __forceinline__ ba_d4 ba_mfma(double a, double b, ba_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ba_f4 ba_mfma(float a, float b, ba_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename T> __device__ __forceinline__ int ba_crow(int lk, int v) { return sizeof(T) == 8 ? lk + 4 * v : 4 * lk + v; }

#endif
