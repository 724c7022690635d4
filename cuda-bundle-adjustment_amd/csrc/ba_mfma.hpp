// ba_mfma.hpp -- the one matrix-core instruction the path uses, in the library's Scalar: v_mfma_f64_16x16x4_f64 (fp64 build) /
// v_mfma_f32_16x16x4_f32 (fp32 build).  Shared by the coarse inversion (ba_coarse.hip) and the exact reduced solve (ba_direct.hip).
#pragma once

#include "ba_device.hpp"

namespace cubahip
{

// Out[i][j] += sum_k A[i][k] B[k][j] on a 16 x 16 x 4 step.  Lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15]; it receives 4
// results of column l & 15, in rows (l >> 4) + 4 q (f64) or 4 (l >> 4) + q (f32), q = 0..3.
#ifdef CUBA_HIP_FLOAT32
typedef float MfmaAcc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MfmaAcc mfma_16x16x4(float a, float b, MfmaAcc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mfma_row(int lane, int q) { return 4 * (lane >> 4) + q; }
#else
typedef double MfmaAcc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MfmaAcc mfma_16x16x4(double a, double b, MfmaAcc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mfma_row(int lane, int q) { return (lane >> 4) + 4 * q; }
#endif
__device__ __forceinline__ MfmaAcc mfma_zero() { return MfmaAcc{ 0, 0, 0, 0 }; }
__device__ __forceinline__ Scalar mfma_get(const MfmaAcc& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

}  // namespace cubahip
