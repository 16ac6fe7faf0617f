// Shared device helpers for the gfx950 (MI355X / CDNA4) anti-spoof kernels.
// Wavefront = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

namespace afx {

// Raising a kernel's dynamic-LDS limit (hipFuncSetAttribute) acts on the CURRENT device's copy of the kernel and is
// sticky there, so the "already raised to N bytes" memo is kept per device (one process may drive several GPUs: the
// reference passes `device=rank`, main.py:48) and under a lock (engines may be driven from several host threads).
// ensure() raises the limit only when a larger request arrives (graph-capture friendly: no call in steady state).
constexpr int kMaxDevices = 16;
struct LdsLimit {
  int have[kMaxDevices] = {0};
  std::mutex mu;
  hipError_t ensure(const void* fn, int bytes) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(mu);
    if (bytes <= have[d]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have[d] = bytes;
    return e;
  }
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
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

// fp32 "exact mode": the operand buffers hold plain floats (no matrix-core intrinsic here:
// the fp32 GEMM and attention are separate kernels, afx_gemm_f32.hip / conf_attn).
struct F32T {
  typedef float T;
  typedef f32x8 V8;
  typedef f32x4 V4;
};

// run `call` with HT bound to the traits of a DType value
#define AFX_DISPATCH_HT(dtype, ...)                               \
  do {                                                            \
    if ((dtype) == DT_BF16) { typedef BF16 HT; __VA_ARGS__; }     \
    else if ((dtype) == DT_FP16) { typedef FP16 HT; __VA_ARGS__; } \
    else { typedef F32T HT; __VA_ARGS__; }                        \
  } while (0)

constexpr float kSeluAlpha = 1.6732632423543772f;
constexpr float kSeluScale = 1.0507009873554805f;

// erf-GELU, 0.5 x (1 + erf(x / sqrt 2)).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7,
// i.e. fp32 rounding level): one v_rcp, one v_exp and 6 fmas instead of the ~40-instruction
// branchy library erff -- the activation sits on 0.4 G elements per utterance of the
// conv stack, where it decided whether those kernels are VALU- or HBM-bound.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164189f, 1.0f));  // 1 / (1 + p |x| / sqrt 2), raw v_rcp (1 ulp)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(ax * ax * -0.7213475204444817f);  // erfc(|x|/sqrt 2) in (0,1]
  return 0.5f * x * (x >= 0.f ? 2.0f - q : q);  // no cancellation on the negative tail
}
// The same on two values at once: everything but the two transcendentals is packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32), which halves the VALU issue slots of the GEMM epilogues.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  const f32x2_t ax = __builtin_elementwise_abs(x);
  const f32x2_t d = __builtin_elementwise_fma(ax, (f32x2_t)(0.23164189f), (f32x2_t)(1.0f));
  const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t)(1.061405429f), (f32x2_t)(-1.453152027f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(1.421413741f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(-0.284496736f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.254829592f));
  const f32x2_t zz = ax * ax * -0.7213475204444817f;
  const f32x2_t e = {__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
  const f32x2_t q = p * t * e;
  const f32x2_t hx = x * 0.5f;
  f32x2_t sel;
  sel[0] = x[0] >= 0.f ? 2.0f - q[0] : q[0];
  sel[1] = x[1] >= 0.f ? 2.0f - q[1] : q[1];
  return hx * sel;
}
// erf-GELU without transcendentals, for results that are rounded to fp16 / bf16 next (|err| <= 2.2e-6 absolute
// over all x in fp32 arithmetic, two orders below the fp16 rounding of the result; gelu_erf2 above stays the
// form of the fp32 "exact mode").  v_rcp / v_exp issue at a quarter of the VALU rate, and the A&S form needs
// two of each per pair: the GELU of the fused conv tile was ~110 of 760 us with that form, 65 with this.  Here
//   u = clamp(x / 5, -1, 1),  s = 2 u^2 - 1 in [-1, 1],  erf(x / sqrt 2) = u (1 + (s - 1) P(s))
// with P of degree 10 fitted (weighted least squares at Chebyshev nodes, weight = the sensitivity of x Phi(x))
// in the well-conditioned variable s -- monomial coefficients <= 0.41, where the plain odd polynomial in x has
// alternating coefficients up to 283 and loses 5e-5 to cancellation in fp32.  The constraint makes erf exactly
// +-1 at the clamp, so gelu(x) = x for x >= 5 and 0 for x <= -5 (true value there: > -1.5e-6).
// 18 packed fp32 ops + 2 v_med3 per pair of values.
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t x) {
  f32x2_t u = x * 0.2f;
  u[0] = __builtin_amdgcn_fmed3f(u[0], -1.0f, 1.0f);
  u[1] = __builtin_amdgcn_fmed3f(u[1], -1.0f, 1.0f);
  const f32x2_t u2 = u * u;
  const f32x2_t s = __builtin_elementwise_fma(u2, (f32x2_t)(2.0f), (f32x2_t)(-1.0f));
  const f32x2_t sm1 = __builtin_elementwise_fma(u2, (f32x2_t)(2.0f), (f32x2_t)(-2.0f));
  f32x2_t p = __builtin_elementwise_fma((f32x2_t)(-1.006885245e-02f), s, (f32x2_t)(2.117710188e-02f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(-1.960056648e-02f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(3.295174241e-02f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(-6.727574021e-02f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(1.008383185e-01f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(-1.358545870e-01f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(1.779851764e-01f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(-2.258825898e-01f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(2.893459797e-01f));
  p = __builtin_elementwise_fma(p, s, (f32x2_t)(-4.136378467e-01f));
  const f32x2_t e = __builtin_elementwise_fma(p, sm1, (f32x2_t)(1.0f)) * u;
  const f32x2_t hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, e, hx);
}
// The same on N values at once (N = 4 or 8), every Horner step written across all of them: left to itself the
// compiler finishes one pair's 13-deep dependent chain (with an s_nop per link) before it starts the next.
template <class V, int N>
__device__ __forceinline__ V gelu_poly_n(V x) {
  V u = x * 0.2f;
#pragma unroll
  for (int i = 0; i < N; ++i) u[i] = __builtin_amdgcn_fmed3f(u[i], -1.0f, 1.0f);
  const V u2 = u * u;
  const V s = __builtin_elementwise_fma(u2, (V)(2.0f), (V)(-1.0f));
  const V sm1 = __builtin_elementwise_fma(u2, (V)(2.0f), (V)(-2.0f));
  V p = __builtin_elementwise_fma((V)(-1.006885245e-02f), s, (V)(2.117710188e-02f));
  p = __builtin_elementwise_fma(p, s, (V)(-1.960056648e-02f));
  p = __builtin_elementwise_fma(p, s, (V)(3.295174241e-02f));
  p = __builtin_elementwise_fma(p, s, (V)(-6.727574021e-02f));
  p = __builtin_elementwise_fma(p, s, (V)(1.008383185e-01f));
  p = __builtin_elementwise_fma(p, s, (V)(-1.358545870e-01f));
  p = __builtin_elementwise_fma(p, s, (V)(1.779851764e-01f));
  p = __builtin_elementwise_fma(p, s, (V)(-2.258825898e-01f));
  p = __builtin_elementwise_fma(p, s, (V)(2.893459797e-01f));
  p = __builtin_elementwise_fma(p, s, (V)(-4.136378467e-01f));
  const V e = __builtin_elementwise_fma(p, sm1, (V)(1.0f)) * u;
  const V hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, e, hx);
}
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 x) { return gelu_poly_n<f32x4, 4>(x); }
__device__ __forceinline__ void gelu_poly8(f32x4& a, f32x4& b) {
  const f32x8 y = gelu_poly_n<f32x8, 8>(f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
  a = f32x4{y[0], y[1], y[2], y[3]};
  b = f32x4{y[4], y[5], y[6], y[7]};
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swish(float x) { return x * sigmoid_acc(x); }
// v_exp + v_rcp forms (about 2 ulp) for results that are rounded to fp16/bf16 next: the accurate
// expf + IEEE division above are ~40 VALU instructions per element, these are 5.  A wave64 VALU
// instruction occupies its SIMD for 4 cycles, so this is what bounds the register-resident
// Conformer chains (one or two waves per SIMD, tools/bench_chain attribution in DESIGN.md).
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float swish_fast(float x) { return x * sigmoid_fast(x); }
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

// Keeps a scalar fp32 chain out of the vectoriser's packed forms (v_pk_fma_f32 ...).  Round 4: conv0_kernel<F32T>,
// compiled with its 10-tap loop packed, returned wrong LOW halves in lanes 48-63 of a channel pair (about one frame in 1e5) whenever an
// fp16 GEMM kernel -- this library's or the vendor's -- started beside it on another stream; compiled with the loop scalar it never did
// (0 of 48 batches beside the vendor GEMM that moved 21 of 24 before).  The packed loop's last writers of those elements are
// `v_pk_fma_f32 v[d:d+1], w, v[d:d+1], acc op_sel:[0,1,0]` (the destination overwrites the sample pair whose HIGH register the LOW result
// reads); isolated in tools/pk_hazard_probe.hip that form alone does NOT fail, so the mechanism is not pinned to it -- the tap loops of the
// VALU conv-layer-0 kernels are scalar regardless, and tools/scan_pk_hazard.py keeps the form out of the built library.
#ifdef AFX_C0_PACKED  // diagnostics only (make variant NAME=c0pk DEFS=-DAFX_C0_PACKED; tools/diag_conv0_pk.py): the loops as hipcc packs them
__device__ __forceinline__ void scalar_only(float&) {}
#else
__device__ __forceinline__ void scalar_only(float& a) { asm volatile("" : "+v"(a)); }
#endif

// Full-wave (64-lane) all-reduce without LDS traffic.  __shfl_xor lowers to ds_bpermute
// (an LDS-crossbar round trip + lgkmcnt wait per step -- six of them per reduction made the
// LayerNorm statistics the most expensive part of conv0/rownorm).  Here: four DPP steps
// inside the 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then
// v_permlane16_swap and v_permlane32_swap fold the four rows -- pure VALU.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// value of the lane 16 (resp. 32) positions away, for every lane
__device__ __forceinline__ float xor16_partner(float v) {
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  // s[0] = rows {0,0,2,2}, s[1] = rows {1,1,3,3} of v: the partner is whichever differs from v
  const float a = __uint_as_float(s[0]), b = __uint_as_float(s[1]);
  return ((threadIdx.x >> 4) & 1) ? a : b;
}
__device__ __forceinline__ float xor32_partner(float v) {
  const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(s[0]), b = __uint_as_float(s[1]);  // a = {lo,lo}, b = {hi,hi}
  return ((threadIdx.x >> 5) & 1) ? a : b;
}
// reduce over the 4 lanes {l, l^16, l^32, l^48} (the 16-lane rows), result in all of them
__device__ __forceinline__ float rows_sum(float v) {
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
  const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// sum over the lane pair {l, l ^ 32} (the two halves that share an output row of a 32x32 MFMA tile)
__device__ __forceinline__ float pair_sum(float v) {
  const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float rows_max(float v) {
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
  const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror: every lane holds its 16-lane row sum
  {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);  // rows {0+1, 0+1, 2+3, 2+3}
  }
  {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
  }
  {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
  }
  return v;
}

}  // namespace afx
