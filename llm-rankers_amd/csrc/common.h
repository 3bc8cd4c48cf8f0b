// Shared device-side types for the gfx950 (CDNA4, wave64) kernels of the reranking engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RK_WAVE 64
// Relative-position bias is looked up through a per-head table indexed by clamp(key-query, -RK_LUT_R, RK_LUT_R).
// Every distance >= max_distance falls in the last bucket (hf: modeling_t5.py:216-262), so the table is exact
// as long as max_distance <= RK_LUT_R (checked at engine creation).
#define RK_LUT_R 128
#define RK_LUT_N (2 * RK_LUT_R + 1)

__device__ __forceinline__ half_t f2h_sat(float x) {
  // fp16 saturation instead of inf (the reference's fp16 path clamps to finfo.max, hf: modeling_t5.py:467-474)
  return (half_t)fminf(fmaxf(x, -65504.f), 65504.f);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// reductions over an aligned group of 16 lanes (one DPP row): quad swaps, then row rotations - every lane ends up with
// the result, no LDS; the order of the additions is fixed
__device__ __forceinline__ float row16_max(float v) {   // max over an aligned group of 16 lanes (DPP, no LDS)
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float row16_sum_f(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

// Folded RMSNorm row factor from the per-block sums of squares a residual GEMM epilogue left: blocks added in increasing
// order, rsqrt(sum / d + eps) / xs.  ONE definition for rowscale_kernel and for the GEMMs that form the factor themselves
// (gemm.h: gemm_row_factors) - which of the two runs depends on the tile variant, i.e. on the batch, so they must agree bit for bit.
__device__ __forceinline__ float rk_row_factor(const float* __restrict__ src, int nb, int d, float eps, float xs) {
  float s = 0.f;
  if ((nb & 3) == 0) {
    for (int j = 0; j < nb; j += 4) { const f32x4 v = *(const f32x4*)(src + j); s += v[0]; s += v[1]; s += v[2]; s += v[3]; }
  } else {
    for (int j = 0; j < nb; ++j) s += src[j];
  }
  return rsqrtf(s / (float)d + eps) / xs;
}
