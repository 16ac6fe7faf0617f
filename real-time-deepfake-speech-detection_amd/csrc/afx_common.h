// Shared device helpers for the gfx950 (MI355X / CDNA4) anti-spoof kernels.
// Wavefront = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Matrix-core operand type traits: bf16 or fp16 (same MFMA rate on gfx950).
struct BF16 {
  typedef __bf16 T;
  typedef bf16x8 V8;
  typedef bf16x4 V4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
struct FP16 {
  typedef _Float16 T;
  typedef f16x8 V8;
  typedef f16x4 V4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr float kSeluAlpha = 1.6732632423543772f;
constexpr float kSeluScale = 1.0507009873554805f;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swish(float x) { return x * sigmoid_acc(x); }
__device__ __forceinline__ float selu(float x) {
  return x > 0.f ? kSeluScale * x : kSeluScale * kSeluAlpha * (expf(x) - 1.0f);
}

// Activation codes shared by the GEMM epilogue and the row kernels.
enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_SWISH = 2, ACT_SELU = 3 };
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_GELU: return gelu_erf(v);
    case ACT_SWISH: return swish(v);
    case ACT_SELU: return selu(v);
    default: return v;
  }
}

// Full-wave (64-lane) butterfly reductions through DPP/permute shuffles.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace afx
