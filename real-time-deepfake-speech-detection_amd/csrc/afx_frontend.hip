// Frontend / row kernels for gfx950: raw-waveform conv layer 0 with fused
// LayerNorm + GELU, generic row LayerNorm, weight packing.
#include <type_traits>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

// ---------------------------------------------------------------------------------
// conv layer 0: Conv1d(1 -> 512, k=10, s=5, bias) ->
// LayerNorm over the 512 channels (fp32 statistics) -> erf-GELU -> operand type.
// HBM-bound by its 1-KB-per-frame output (256 KB of waveform in, 13 MB out per 4-s
// utterance).  One WAVE per frame: lane l owns channels 8l..8l+7 (weights, bias,
// gamma, beta resident in VGPRs for the whole block), the 10 input samples are LDS
// broadcasts, LayerNorm statistics are two 64-lane shuffle reductions, and the wave
// stores one contiguous 1-KB row (16 B per lane).  Optional pre-emphasis
// (data/preprocess.py:16-29) is applied while the waveform window is staged in LDS.
// ---------------------------------------------------------------------------------
constexpr int C0_FB = 64;  // frames per workgroup
static int g_conv0_mfma = 1;  // A/B knob: 1 = matrix-core forms of conv layer 0 (split-precision fp16 with a packed operand, else fp32 MFMA), 2 = fp32 MFMA always, 0 = the VALU form below

template <class HT>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wave, int L, int T0,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int pre_emph, float pre_coef, typename HT::T* __restrict__ out) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  __shared__ float xs[C0_FB * 5 + 8];
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * C0_FB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* x = wave + (long)b * L;
  const int s0 = f0 * 5;
  for (int i = tid; i < C0_FB * 5 + 5; i += 256) {
    const int gidx = s0 + i;
    float v = 0.f;
    if (gidx < L) {
      v = x[gidx];
      if (pre_emph) {
        const int gp = gidx > 0 ? gidx - 1 : 1;  // reflect pad of one sample on the left
        v -= pre_coef * x[gp];
      }
    }
    xs[i] = v;
  }
  float wr[8][10], bi[8], ga[8], be[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane * 8 + i;
#pragma unroll
    for (int j = 0; j < 10; ++j) wr[i][j] = w[c * 10 + j];
    bi[i] = bias[c];
    ga[i] = gamma[c];
    be[i] = beta[c];
  }
  __syncthreads();
  const int fend = min(C0_FB, T0 - f0);
  for (int f = wv; f < fend; f += 4) {
    float xv[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) xv[j] = xs[f * 5 + j];
    float v[8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = bi[i];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        a = fmaf(wr[i][j], xv[j], a);
        scalar_only(a);
      }
      v[i] = a;
      sum += a;
    }
#ifdef AFX_C0_KEEPX  // diagnostics only (with AFX_C0_PACKED): the samples stay live past the loop, so the packed loop cannot write
#pragma unroll       // an accumulator over the sample pair it reads -- packed math WITHOUT the in-place cross-half form
    for (int j = 0; j < 10; ++j) asm volatile("" ::"v"(xv[j]), "v"(sum));  // (after the last accumulation: it needs `sum`)
#endif
    const float mean = wave_sum(sum) * (1.0f / 512.0f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] -= mean;
      sq = fmaf(v[i], v[i], sq);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / 512.0f) + 1e-5f);
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {  // two channels per packed-math GELU
      const f32x2_t yin = f32x2_t{fmaf(v[i] * rstd, ga[i], be[i]), fmaf(v[i + 1] * rstd, ga[i + 1], be[i + 1])};
      const f32x2_t y = sizeof(T) == 4 ? gelu_erf2(yin) : gelu_poly2(yin);  // fp32 = exact mode
      o[i] = (T)y[0];
      o[i + 1] = (T)y[1];
    }
    *(V8*)(out + ((long)b * T0 + f0 + f) * 512 + lane * 8) = o;
  }
}

// The same layer with the 10-tap products on the fp32 matrix instruction (exact fp32, like the VALU
// form above): out^T tile = W (16 channels x 12 taps, zero padded) . frames (12 x 16), three
// v_mfma_f32_16x16x4_f32 per 16-channel tile.  A lane then holds 4 consecutive channels x 32 tiles
// of ONE frame: the LayerNorm statistics are a 4-lane reduction per 16 frames (not a 64-lane one per
// frame), and only LayerNorm + GELU + convert remain on the VALU: ~80 instructions per frame instead
// of ~185.  Workgroup = 256 frames (4 waves x 4 groups of 16); weights (24 KB, tap-padded) and the
// waveform window live in LDS.
#ifndef CONV0_DBG
#define CONV0_DBG 0  // timing experiments only (wrong results): 1 no GELU, 2 no stores, 4 no MFMA, 8 no LayerNorm statistics
#endif
constexpr int C0M_FB = 256;
template <class HT>
__global__ __launch_bounds__(256, 2) void conv0_mfma_kernel(const float* __restrict__ wave, int L, int T0,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int pre_emph, float pre_coef, typename HT::T* __restrict__ out) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  __shared__ float xs[C0M_FB * 5 + 16];
  __shared__ float ws[3 * 32 * 64];      // [k-step][channel tile][lane]: the lane's A-operand value
  __shared__ float pv[3 * 512];          // bias | gamma | beta
  const int b = blockIdx.y, f0 = blockIdx.x * C0M_FB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* x = wave + (long)b * L;
  const int s0 = f0 * 5;
  for (int i = tid; i < C0M_FB * 5 + 16; i += 256) {
    const int gidx = s0 + i;
    float v = 0.f;
    if (gidx < L) {
      v = x[gidx];
      if (pre_emph) {
        const int gp = gidx > 0 ? gidx - 1 : 1;  // reflect pad of one sample on the left
        v -= pre_coef * x[gp];
      }
    }
    xs[i] = v;
  }
  for (int i = tid; i < 3 * 32 * 64; i += 256) {
    const int s = i / 2048, ct = (i >> 6) & 31, l = i & 63;
    const int tap = 4 * s + (l >> 4);
    ws[i] = tap < 10 ? w[(ct * 16 + (l & 15)) * 10 + tap] : 0.f;
  }
  for (int i = tid; i < 512; i += 256) {
    pv[i] = bias[i];
    pv[512 + i] = gamma[i];
    pv[1024 + i] = beta[i];
  }
  __syncthreads();
  const int fr = lane & 15, kq = lane >> 4;
  for (int grp = 0; grp < 4; ++grp) {
    const int fl = (wv * 4 + grp) * 16;  // first frame of this group inside the workgroup
    if (f0 + fl >= T0) break;
    float bx[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) bx[s] = xs[(fl + fr) * 5 + 4 * s + kq];
    f32x4 acc[32];
#pragma unroll
    for (int ct = 0; ct < 32; ++ct) {
      f32x4 c = *(const f32x4*)(pv + ct * 16 + kq * 4);  // bias: the accumulator's own channels
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if constexpr ((CONV0_DBG & 4) != 0) c[0] += ws[(s * 32 + ct) * 64 + lane] * bx[s];
        else c = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[(s * 32 + ct) * 64 + lane], bx[s], c, 0, 0, 0);
      }
      acc[ct] = c;
    }
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 32; ++ct) sum += (acc[ct][0] + acc[ct][1]) + (acc[ct][2] + acc[ct][3]);
    const float mean = rows_sum(sum) * (1.0f / 512.0f);
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < 32; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[ct][r] -= mean;
        sq = fmaf(acc[ct][r], acc[ct][r], sq);
      }
    const float rstd = 1.0f / sqrtf(rows_sum(sq) * (1.0f / 512.0f) + 1e-5f);
    const int f = f0 + fl + fr;
    T* orow = out + ((long)b * T0 + (f < T0 ? f : T0 - 1)) * 512;
    const int cb = (kq & 1) * 16 + (kq >> 1) * 8;
#pragma unroll
    for (int cp = 0; cp < 16; ++cp) {  // channel-tile pairs -> 8 consecutive channels per lane
      f32x4 va, vb;
      {
        const f32x4 g0 = *(const f32x4*)(pv + 512 + cp * 32 + kq * 4), b0 = *(const f32x4*)(pv + 1024 + cp * 32 + kq * 4);
        const f32x4 g1 = *(const f32x4*)(pv + 512 + cp * 32 + 16 + kq * 4), b1 = *(const f32x4*)(pv + 1024 + cp * 32 + 16 + kq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          va[r] = fmaf(acc[2 * cp][r] * rstd, g0[r], b0[r]);
          vb[r] = fmaf(acc[2 * cp + 1][r] * rstd, g1[r], b1[r]);
        }
        if constexpr ((CONV0_DBG & 1) == 0) gelu_poly8(va, vb);  // all 8 values step by step together
      }
      V8 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va[r]), __float_as_uint(vb[r]), false, false);
        h[r] = (T)__uint_as_float(sw[0]);
        h[4 + r] = (T)__uint_as_float(sw[1]);
      }
      if constexpr ((CONV0_DBG & 2) != 0) asm volatile("" :: "v"(h));
      else if (f < T0) *(V8*)(orow + cp * 32 + cb) = h;
    }
  }
}

// ---------------------------------------------------------------------------------
// conv layer 0 at fp32 accuracy on the fp16 matrix pipe (the half-precision engines' form): the 10 taps, their
// split-precision corrections and the bias are ONE v_mfma_f32_16x16x32_f16 per 16 channels x 16 frames instead of
// three v_mfma_f32_16x16x4_f32 (16 against 96 matrix-pipe cycles):
//     k  0.. 9  xh[tap] * wh[tap]          x = xh + xl, w = wh + wl (fp16 hi + fp16 lo of the remainder)
//     k 10..19  xl[tap] * wh[tap]
//     k 20..29  xh[tap] * wl[tap]          (xl * wl, 2^-22 relative, is dropped)
//     k 30, 31  (G / cb) * (cb bias)_hi, (G / cb) * (cb bias)_lo      (cb: power of two, largest |bias| into [1, 2))
// Each frame is scaled by a power of two S_f that brings its largest sample into [1, 2) and the weights by one power of
// two c for the layer (largest |w| into [1, 2)), so that the lo parts sit in fp16's normal range whatever the recording
// level; G = S_f * c multiplies the whole pre-norm row and the LayerNorm that follows divides it out again
// (eps -> G^2 eps).  S_f depends on the frame's own 10 samples only: a frame's result does not depend on its
// neighbours, the batch or the launch (the streaming scorer relies on that).
// The packed weight operand (conv0_pack_kernel, once per checkpoint): [32 channel tiles][64 lanes][8 halfs] in MFMA
// A-operand order, followed by the exponents of c and cb.
// ---------------------------------------------------------------------------------
constexpr int C0P_HALFS = 32 * 64 * 8;
__global__ __launch_bounds__(256) void conv0_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                         _Float16* __restrict__ pack) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  auto norm_exp = [&](const float* v, int n) {  // e with max|v| * 2^e in [1, 2), clamped to +-12
    float m = 0.f;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, fabsf(v[i]));
    __syncthreads();
    red[tid] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
      __syncthreads();
    }
    const float mx = red[0];
    int e = 0;
    if (mx > 0.f && mx < INFINITY) e = 127 - (int)((__float_as_uint(mx) >> 23) & 0xff);
    return e < -12 ? -12 : (e > 12 ? 12 : e);
  };
  const int cexp = norm_exp(w, 512 * 10), bexp = norm_exp(bias, 512);
  const float c = ldexpf(1.0f, cexp), cb = ldexpf(1.0f, bexp);
  for (int i = tid; i < 32 * 64; i += 256) {
    const int ct = i >> 6, l = i & 63, ch = ct * 16 + (l & 15), kg = l >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kg * 8 + j;
      float v;
      if (k < 30) {
        const float wv = w[ch * 10 + (k < 10 ? k : (k < 20 ? k - 10 : k - 20))] * c;
        const float hi = (float)(_Float16)wv;
        v = k < 20 ? hi : wv - hi;
      } else {
        const float bv = bias[ch] * cb;
        const float hi = (float)(_Float16)bv;
        v = k == 30 ? hi : bv - hi;
      }
      pack[i * 8 + j] = (_Float16)v;
    }
  }
  if (tid == 0) {
    ((int*)(pack + C0P_HALFS))[0] = cexp;
    ((int*)(pack + C0P_HALFS))[1] = bexp;
  }
}

// HT = F32T (round 4): the split-precision engines ("fp16x3") run this kernel too -- its products are hi / lo fp16 pairs already --
// with the fp32-accurate erf-GELU and fp32 rows out, or, pair_scale > 0, the rows leave as conv layer 1's PAIR-FORM operand
// (afx_kernels.h) scaled by that power of two: no fp32 round trip and no split launch over the stack's largest activation.
template <class HT>
__global__ __launch_bounds__(256, 2) void conv0_split_kernel(const float* __restrict__ wave, int L, int T0,
                                                          const _Float16* __restrict__ pack, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int pre_emph, float pre_coef,
                                                          typename HT::T* __restrict__ out, float pair_scale) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  constexpr bool F32 = std::is_same<T, float>::value;
  __shared__ float xs[C0M_FB * 5 + 16];
  __shared__ __attribute__((aligned(16))) _Float16 xop[C0M_FB * 32];  // per frame: the 32 k-values of its B operand
  __shared__ float Gs[C0M_FB];
  __shared__ __attribute__((aligned(16))) _Float16 wsh[C0P_HALFS];
  __shared__ float pv[2 * 512];  // gamma | beta
  const int b = blockIdx.y, f0 = blockIdx.x * C0M_FB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* x = wave + (long)b * L;
  const int s0 = f0 * 5;
  for (int i = tid; i < C0M_FB * 5 + 16; i += 256) {
    const int gidx = s0 + i;
    float v = 0.f;
    if (gidx < L) {
      v = x[gidx];
      if (pre_emph) {
        const int gp = gidx > 0 ? gidx - 1 : 1;  // reflect pad of one sample on the left
        v -= pre_coef * x[gp];
      }
    }
    xs[i] = v;
  }
  for (int i = tid; i < C0P_HALFS / 8; i += 256) ((uint4*)wsh)[i] = ((const uint4*)pack)[i];
  const int cexp = ((const int*)(pack + C0P_HALFS))[0], bexp = ((const int*)(pack + C0P_HALFS))[1];
  for (int i = tid; i < 512; i += 256) {
    pv[i] = gamma[i];
    pv[512 + i] = beta[i];
  }
  __syncthreads();
  {  // thread = frame: scale, split, lay out the 32 k-values
    float xv[10], m = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      xv[j] = xs[tid * 5 + j];
      m = fmaxf(m, fabsf(xv[j]));
    }
    const int eb = (int)((__float_as_uint(m) >> 23) & 0xff);
    int ge = (eb > 0 && eb < 255 ? 127 - eb : 0) + cexp;  // exponent of G = S_f * c; S_f * m in [1, 2)
    int qe = ge - bexp;                                   // G / cb is an fp16 operand (k = 30, 31): keep it normal
    qe = qe < -14 ? -14 : (qe > 15 ? 15 : qe);
    ge = qe + bexp;
    // a frame far beyond any audio scale (|x| > 2^14 / its scale): keep S_f * m inside fp16 and let the bias operand
    // underflow instead -- against such a signal the bias is below fp32 resolution anyway
    if (eb > 0 && eb < 255 && ge - cexp + (eb - 127) > 14) {
      ge = 14 - (eb - 127) + cexp;
      qe = ge - bexp;
    }
    const float S = ldexpf(1.0f, ge - cexp), G = ldexpf(1.0f, ge);
    _Float16 h[32];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const float v = xv[j] * S;
      const _Float16 hi = (_Float16)v;
      h[j] = hi;
      h[10 + j] = (_Float16)(v - (float)hi);
      h[20 + j] = hi;
    }
    h[30] = h[31] = (_Float16)ldexpf(1.0f, qe);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = h[q * 8 + j];
      *(f16x8*)(xop + tid * 32 + q * 8) = t;
    }
    Gs[tid] = G;
  }
  __syncthreads();
  const int fr = lane & 15, kq = lane >> 4;
  for (int grp = 0; grp < 4; ++grp) {
    const int fl = (wv * 4 + grp) * 16;  // first frame of this group inside the workgroup
    if (f0 + fl >= T0) break;
    const f16x8 bx = *(const f16x8*)(xop + (fl + fr) * 32 + kq * 8);
    const float G = Gs[fl + fr];
    f32x4 acc[32];
#pragma unroll
    for (int ct = 0; ct < 32; ++ct)
      acc[ct] = FP16::mfma(*(const f16x8*)(wsh + (ct * 64 + lane) * 8), bx, f32x4{0.f, 0.f, 0.f, 0.f});
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 32; ++ct) sum += (acc[ct][0] + acc[ct][1]) + (acc[ct][2] + acc[ct][3]);
    const float mean = rows_sum(sum) * (1.0f / 512.0f);
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < 32; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[ct][r] -= mean;
        sq = fmaf(acc[ct][r], acc[ct][r], sq);
      }
    const float rstd = 1.0f / sqrtf(rows_sum(sq) * (1.0f / 512.0f) + (G * G) * 1e-5f);
    const int f = f0 + fl + fr;
    T* orow = out + ((long)b * T0 + (f < T0 ? f : T0 - 1)) * 512;
    const int cb = (kq & 1) * 16 + (kq >> 1) * 8;
#pragma unroll
    for (int cp = 0; cp < 16; ++cp) {  // channel-tile pairs -> 8 consecutive channels per lane
      f32x4 va, vb;
      {
        const f32x4 g0 = *(const f32x4*)(pv + cp * 32 + kq * 4), b0 = *(const f32x4*)(pv + 512 + cp * 32 + kq * 4);
        const f32x4 g1 = *(const f32x4*)(pv + cp * 32 + 16 + kq * 4), b1 = *(const f32x4*)(pv + 512 + cp * 32 + 16 + kq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          va[r] = fmaf(acc[2 * cp][r] * rstd, g0[r], b0[r]);
          vb[r] = fmaf(acc[2 * cp + 1][r] * rstd, g1[r], b1[r]);
        }
        if constexpr (F32) {  // fp32 results: the fp32-accurate erf form (the polynomial is sized for fp16 outputs)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            va[r] = gelu_erf(va[r]);
            vb[r] = gelu_erf(vb[r]);
          }
        } else {
          gelu_poly8(va, vb);
        }
      }
      float w8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va[r]), __float_as_uint(vb[r]), false, false);
        w8[r] = __uint_as_float(sw[0]);
        w8[4 + r] = __uint_as_float(sw[1]);
      }
      if (f >= T0) continue;
      if constexpr (F32) {
        if (pair_scale > 0.f) {  // columns cp * 32 + cb .. + 7 of the row's pair form: inside one 32-element group
          f16x8 hi, lo;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float sv = w8[r] * pair_scale;
            hi[r] = (_Float16)sv;
            lo[r] = (_Float16)(sv - (float)hi[r]);
          }
          _Float16* hp = (_Float16*)orow + cp * 64 + cb;
          *(f16x8*)hp = hi;
          *(f16x8*)(hp + 32) = lo;
          continue;
        }
      }
      V8 h;
#pragma unroll
      for (int r = 0; r < 8; ++r) h[r] = (T)w8[r];
      *(V8*)(orow + cp * 32 + cb) = h;
    }
  }
}

size_t conv0_pack_bytes() { return (size_t)C0P_HALFS * 2 + 16; }
const char* launch_conv0_pack(const float* w, const float* bias, void* pack, hipStream_t s) {
  if (!w || !bias || !pack) return "conv0_pack: null argument";
  hipLaunchKernelGGL(conv0_pack_kernel, dim3(1), dim3(256), 0, s, w, bias, (_Float16*)pack);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// conv layer 0 of the wav2vec2-*base* feature extractor (fairseq extractor_mode="default"; SURVEY 8a row 1a, the
// "GroupNorm/GELU" `north_star` names): bias-free Conv1d(1 -> 512, k=10, s=5) -> GroupNorm(512 groups of one channel
// = per (utterance, channel) normalisation over TIME, affine) -> erf-GELU.  The statistics span the whole clip, so
// the layer is two passes over the (cheap: 10 MACs per output) convolution: pass 1 writes per-chunk partial sums
// (sum, sum of squares) -- fixed chunks, summed in a fixed order in double by the finalize kernel, so the result does
// not depend on scheduling --, pass 2 recomputes the convolution, normalises, activates and stores.
// ---------------------------------------------------------------------------------
constexpr int GN_FB = 256;  // frames per statistics chunk
__global__ __launch_bounds__(256) void conv0_gn_stats_kernel(const float* __restrict__ wave, int L, int T0,
                                                             const float* __restrict__ w, float* __restrict__ part /*[B][nchunk][2][512]*/) {
  __shared__ float xs[GN_FB * 5 + 8];
  const int b = blockIdx.y, f0 = blockIdx.x * GN_FB, tid = threadIdx.x;
  const float* x = wave + (long)b * L;
  for (int i = tid; i < GN_FB * 5 + 5; i += 256) xs[i] = f0 * 5 + i < L ? x[f0 * 5 + i] : 0.f;
  float w0[10], w1[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    w0[j] = w[tid * 10 + j];
    w1[j] = w[(tid + 256) * 10 + j];
  }
  __syncthreads();
  const int nf = min(GN_FB, T0 - f0);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  for (int f = 0; f < nf; ++f) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const float xv = xs[f * 5 + j];
      a0 = fmaf(w0[j], xv, a0);
      a1 = fmaf(w1[j], xv, a1);
      scalar_only(a0);
      scalar_only(a1);
    }
    s0 += a0; q0 = fmaf(a0, a0, q0);
    s1 += a1; q1 = fmaf(a1, a1, q1);
  }
  float* o = part + ((long)b * gridDim.x + blockIdx.x) * 1024;
  o[tid] = s0; o[tid + 256] = s1; o[512 + tid] = q0; o[512 + tid + 256] = q1;
}
__global__ void conv0_gn_finalize_kernel(const float* __restrict__ part, int nchunk, int T0, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd) {
  const int b = blockIdx.x, c = threadIdx.x;  // 512 threads
  double s = 0.0, q = 0.0;
  for (int k = 0; k < nchunk; ++k) {
    const float* o = part + ((long)b * nchunk + k) * 1024;
    s += (double)o[c];
    q += (double)o[512 + c];
  }
  const double m = s / T0, v = q / T0 - m * m;
  mean[b * 512 + c] = (float)m;
  rstd[b * 512 + c] = (float)(1.0 / sqrt((v > 0.0 ? v : 0.0) + (double)eps));
}
template <class HT>
__global__ __launch_bounds__(256) void conv0_gn_apply_kernel(const float* __restrict__ wave, int L, int T0,
                                                             const float* __restrict__ w, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, typename HT::T* __restrict__ out) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  __shared__ float xs[C0_FB * 5 + 8];
  const int b = blockIdx.y, f0 = blockIdx.x * C0_FB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* x = wave + (long)b * L;
  for (int i = tid; i < C0_FB * 5 + 5; i += 256) xs[i] = f0 * 5 + i < L ? x[f0 * 5 + i] : 0.f;
  float wr[8][10], sc[8], sh[8];  // lane owns channels 8 lane .. 8 lane + 7; y = conv * sc + sh
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane * 8 + i;
#pragma unroll
    for (int j = 0; j < 10; ++j) wr[i][j] = w[c * 10 + j];
    const float r = rstd[b * 512 + c] * gamma[c];
    sc[i] = r;
    sh[i] = fmaf(-mean[b * 512 + c], r, beta[c]);
  }
  __syncthreads();
  const int fend = min(C0_FB, T0 - f0);
  for (int f = wv; f < fend; f += 4) {
    float xv[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) xv[j] = xs[f * 5 + j];
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        a0 = fmaf(wr[i][j], xv[j], a0);
        a1 = fmaf(wr[i + 1][j], xv[j], a1);
        scalar_only(a0);
        scalar_only(a1);
      }
      const f32x2_t yin = f32x2_t{fmaf(a0, sc[i], sh[i]), fmaf(a1, sc[i + 1], sh[i + 1])};
      const f32x2_t y = sizeof(T) == 4 ? gelu_erf2(yin) : gelu_poly2(yin);
      o[i] = (T)y[0];
      o[i + 1] = (T)y[1];
    }
    *(V8*)(out + ((long)b * T0 + f0 + f) * 512 + lane * 8) = o;
  }
}
const char* launch_conv0_groupnorm(const float* wave, int B, int L, int T0, const float* w, const float* gamma,
                                   const float* beta, float eps, float* stats /* B*(nchunk*1024 + 1024) floats */, void* out_h,
                                   int dtype, hipStream_t s) {
  if (B <= 0 || L < 10 || T0 != (L - 10) / 5 + 1 || B > 65535) return "conv0 (group norm): bad shape";
  const int nchunk = (T0 + GN_FB - 1) / GN_FB;
  float* part = stats;
  float* mean = stats + (size_t)B * nchunk * 1024;
  float* rstd = mean + (size_t)B * 512;
  hipLaunchKernelGGL(conv0_gn_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, wave, L, T0, w, part);
  hipLaunchKernelGGL(conv0_gn_finalize_kernel, dim3(B), dim3(512), 0, s, part, nchunk, T0, eps, mean, rstd);
  dim3 grid((T0 + C0_FB - 1) / C0_FB, B);
  AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL(conv0_gn_apply_kernel<HT>, grid, dim3(256), 0, s, wave, L, T0, w, mean, rstd, gamma,
                                            beta, (HT::T*)out_h));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
size_t conv0_groupnorm_stats_floats(int B, int T0) { return (size_t)B * (((size_t)T0 + GN_FB - 1) / GN_FB * 1024 + 1024); }

const char* launch_conv0(const float* wave, int B, int L, int T0, const float* w, const float* bias,
                         const float* gamma, const float* beta, int pre_emph, float pre_coef, void* out_h,
                         int dtype, hipStream_t s, const void* wpack, float pair_scale) {
  if (B <= 0 || L < 10 || T0 != (L - 10) / 5 + 1) return "conv0: bad shape";
  if (dtype == DT_FP16X3) {  // the split-precision engines: fp32 rows (or conv layer 1's pair-form operand) out
    if (!wpack) return "conv0 (fp16x3): the packed operand block is missing";
    dim3 grid((T0 + C0M_FB - 1) / C0M_FB, B);
    hipLaunchKernelGGL(conv0_split_kernel<F32T>, grid, dim3(256), 0, s, wave, L, T0, (const _Float16*)wpack, gamma, beta, pre_emph,
                       pre_coef, (float*)out_h, pair_scale);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
  }
  if (dtype != DT_FP32 && g_conv0_mfma == 1 && wpack) {  // split precision on the fp16 matrix pipe (packed operand given)
    dim3 grid((T0 + C0M_FB - 1) / C0M_FB, B);
    if (dtype == DT_BF16)
      hipLaunchKernelGGL(conv0_split_kernel<BF16>, grid, dim3(256), 0, s, wave, L, T0, (const _Float16*)wpack, gamma, beta,
                         pre_emph, pre_coef, (__bf16*)out_h, 0.f);
    else
      hipLaunchKernelGGL(conv0_split_kernel<FP16>, grid, dim3(256), 0, s, wave, L, T0, (const _Float16*)wpack, gamma, beta,
                         pre_emph, pre_coef, (_Float16*)out_h, 0.f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
  }
  if (dtype != DT_FP32 && g_conv0_mfma) {
    dim3 grid((T0 + C0M_FB - 1) / C0M_FB, B);
    if (dtype == DT_BF16)
      hipLaunchKernelGGL(conv0_mfma_kernel<BF16>, grid, dim3(256), 0, s, wave, L, T0, w, bias, gamma, beta, pre_emph,
                         pre_coef, (__bf16*)out_h);
    else
      hipLaunchKernelGGL(conv0_mfma_kernel<FP16>, grid, dim3(256), 0, s, wave, L, T0, w, bias, gamma, beta, pre_emph,
                         pre_coef, (_Float16*)out_h);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
  }
  dim3 grid((T0 + C0_FB - 1) / C0_FB, B);
  AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL(conv0_kernel<HT>, grid, dim3(256), 0, s, wave, L, T0, w, bias, gamma, beta,
                                            pre_emph, pre_coef, (HT::T*)out_h));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
void conv0_set_mfma(int v) { g_conv0_mfma = v; }

// ---------------------------------------------------------------------------------
// Stand-alone pre-emphasis (data/preprocess.py:16-29) for callers that apply it as a
// separate module (trainer.py:104); the engine itself fuses it into conv0.
// ---------------------------------------------------------------------------------
__global__ void pre_emphasis_kernel(const float* __restrict__ x, int L, float coef, float* __restrict__ y) {
  const long base = (long)blockIdx.y * L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
    const int p = i > 0 ? i - 1 : (L > 1 ? 1 : 0);  // reflect pad of one sample on the left
    y[base + i] = x[base + i] - coef * x[base + p];
  }
}
const char* launch_pre_emphasis(const float* x, int B, int L, float coef, float* y, hipStream_t s) {
  if (B <= 0 || L <= 0) return "pre_emphasis: empty input";
  hipLaunchKernelGGL(pre_emphasis_kernel, dim3(min((L + 255) / 256, 1024), B), dim3(256), 0, s, x, L, coef, y);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// Utterance length policy as one batched device op (data/test_set.py:139-146 `pad`,
// :201-227 `adjustDuration`, :229-248 `adjustDuration_random_start`): all three are
//     out[b][i] = x_b[(start_b + i) mod n_b],   i < duration
// (a short clip is repeated whole plus a residue, a long one is cropped from start_b; start_b = 0
// for the first-N policies).  x is the ragged batch packed back to back, offs[b] .. offs[b+1] its
// sample range.
// ---------------------------------------------------------------------------------
__global__ void tile_crop_kernel(const float* __restrict__ x, const long long* __restrict__ offs,
                                 const long long* __restrict__ starts, int duration, float* __restrict__ out) {
  const int b = blockIdx.y;
  const long long o = offs[b], n = offs[b + 1] - o;
  long long p = starts ? starts[b] : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < duration; i += gridDim.x * blockDim.x)
    out[(long)b * duration + i] = x[o + (p + i) % n];
}
const char* launch_tile_crop(const float* x, const long long* offs, const long long* starts, int B, int duration,
                             float* out, hipStream_t s) {
  if (B <= 0 || duration <= 0) return "tile_crop: empty batch";
  hipLaunchKernelGGL(tile_crop_kernel, dim3(min((duration + 255) / 256, 256), B), dim3(256), 0, s, x, offs, starts,
                     duration, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// Row LayerNorm (+ activation): one wave per row, C <= 1024, C % 4 == 0.  The row
// stays in registers (float4 per lane per 256-column slab), two-pass statistics in
// fp32 like torch.  Used for the conv-stack LayerNorm+GELU, every transformer /
// Conformer LayerNorm and the final encoder LayerNorm.  HBM-bound.
// ---------------------------------------------------------------------------------
template <class HT, int RPW>  // RPW rows per wave: all their loads are in flight before any is reduced
__global__ __launch_bounds__(256) void rownorm_kernel(RowNormArgs a) {
  typedef typename HT::T T;
  typedef typename HT::V4 V4;
  const int lane = threadIdx.x & 63;
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (r0 >= a.rows) return;
  f32x4 v[RPW][4];
  bool live[RPW];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    live[u] = r0 + u < a.rows;
    const float* x = a.x + (live[u] ? r0 + u : r0) * a.ldx;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = (it * 64 + lane) * 4;
      v[u][it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < a.C) v[u][it] = *(const f32x4*)(x + c);
    }
  }
  // gamma / beta once per wave, requested before any store: on gfx9 a wait for a load that was issued after a
  // store is a wait for the store as well (one shared, out-of-order vmcnt)
  f32x4 gm[4], bt[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = (it * 64 + lane) * 4;
    gm[it] = bt[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < a.C) {
      gm[it] = *(const f32x4*)(a.gamma + c);
      bt[it] = *(const f32x4*)(a.beta + c);
    }
  }
  const float inv = 1.0f / (float)a.C;
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    if (!live[u]) continue;  // wave-uniform
    const long r = r0 + u;
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) sum += (v[u][it][0] + v[u][it][1]) + (v[u][it][2] + v[u][it][3]);
    const float mean = wave_sum(sum) * inv;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = (it * 64 + lane) * 4;
      if (c < a.C) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[u][it][i] -= mean;
          sq = fmaf(v[u][it][i], v[u][it][i], sq);
        }
      }
    }
    const float var = wave_sum(sq) * inv;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    // overflow guard: an operand copy that left fp16's range upstream arrives here as inf / NaN statistics (NaN fails every compare)
    if (a.nonfinite && lane == 0 && !(fabsf(mean) <= 3.0e38f && var <= 3.0e38f)) atomicAdd(a.nonfinite, 1);
    const long orow = (r / a.rpb) * a.o_batch_rows + (r % a.rpb) + a.o_row_off;
    // the activation branch sits OUTSIDE the element loops: inlined per element, the three activations were
    // 3 k of this kernel's 3.8 k instructions (every LayerNorm of the two models is activation-free)
    auto emit = [&](auto with_act) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < a.C) {
          const f32x4 g = gm[it], b = bt[it];
          f32x4 y;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            y[i] = fmaf(v[u][it][i] * rstd, g[i], b[i]);
            if (decltype(with_act)::value) y[i] = apply_act(y[i], a.act);
          }
          if (a.out_f) *(f32x4*)(a.out_f + orow * a.ldo_f + c) = y;
          if (a.out_h) {
            if (std::is_same<T, float>::value && a.oh_pairs) {  // split precision: the next product's A operand, pair form
              f16x4 hi, lo;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float sv = y[i] * a.oh_scale;
                hi[i] = (_Float16)sv;
                lo[i] = (_Float16)(sv - (float)hi[i]);
              }
              _Float16* hp = (_Float16*)a.out_h + orow * (2 * a.ldo_h) + s3_pair_index(c);
              *(f16x4*)hp = hi;
              *(f16x4*)(hp + 32) = lo;
            } else {
              V4 h;
#pragma unroll
              for (int i = 0; i < 4; ++i) h[i] = (T)y[i];
              *(V4*)((T*)a.out_h + orow * a.ldo_h + c) = h;
            }
          }
        }
      }
    };
    if (a.act == ACT_NONE) emit(std::false_type{});
    else emit(std::true_type{});
  }
}

const char* launch_rownorm(const RowNormArgs& a, int dtype, hipStream_t s) {
  if (a.rows <= 0 || a.C <= 0 || a.C > 1024 || a.C % 4) return "rownorm: need 0 < C <= 1024, C % 4 == 0";
  if (!a.out_f && !a.out_h) return "rownorm: no output";
  // two rows per wave once there are enough rows to keep 8+ waves on every CU; below that (the teacher's
  // 16 x 199 rows) one row per wave = twice the waves in flight, which is what this latency-bound size lacks
  if (a.rows >= 16384) {
    AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL((rownorm_kernel<HT, 2>), dim3((a.rows + 7) / 8), dim3(256), 0, s, a));
  } else {
    AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL((rownorm_kernel<HT, 1>), dim3((a.rows + 3) / 4), dim3(256), 0, s, a));
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// zero the time-padding rows of the positional-conv operand buffer (B, pf+T+pb, C)
// ---------------------------------------------------------------------------------
// (C counts 2-byte units: an fp32 buffer passes 2 x its channel count)
__global__ void zero_pad_rows_kernel(uint16_t* buf, int T, int C, int pf, int pb, const int* __restrict__ lens) {
  const int b = blockIdx.y;
  const int keep = lens ? lens[b] : T;  // ragged batch: the frames past this utterance's own length count as padding
  const int rows = pf + pb + (T - keep);
  const long per = (long)(pf + T + pb) * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)rows * C / 8; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8;
    int r = (int)(e / C);
    const int c = (int)(e % C);
    if (r >= pf) r += keep;
    *(u32x4*)(buf + b * per + (long)r * C + c) = u32x4{0u, 0u, 0u, 0u};
  }
}
const char* launch_zero_pad_rows(void* buf_h, int B, int T, int C, int pf, int pb, int dtype, hipStream_t s, const int* lens) {
  if (C % 8) return "zero_pad_rows: C % 8 != 0";
  if (dtype == DT_FP32) C *= 2;
  hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(32, B), dim3(256), 0, s, (uint16_t*)buf_h, T, C, pf, pb, lens);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// weight packing (one-off): fp32 checkpoint layouts -> K-contiguous operand rows
// ---------------------------------------------------------------------------------
template <class HT>
__global__ void pack_linear_kernel(const float* w, int N, int K, int Kpad, typename HT::T* out) {
  const long total = (long)N * Kpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / Kpad), k = (int)(i % Kpad);
    out[i] = (typename HT::T)(k < K ? w[(long)n * K + k] : 0.f);
  }
}
const char* launch_pack_linear(const float* w, int N, int K, int Kpad, void* out_h, int dtype, hipStream_t s) {
  const int blocks = (int)min((long)4096, ((long)N * Kpad + 255) / 256);
  AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL(pack_linear_kernel<HT>, dim3(blocks), dim3(256), 0, s, w, N, K, Kpad,
                                            (HT::T*)out_h));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// Conv1d weight [N][Cin][k] -> [N][j*Cin + c]: tap-major K so that, with channel-last
// activations, a conv row is one contiguous slice of the input.
template <class HT>
__global__ void pack_conv_kernel(const float* w, int N, int Cin, int k, typename HT::T* out) {
  const long total = (long)N * Cin * k;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / ((long)Cin * k));
    const int rem = (int)(i % ((long)Cin * k));
    const int j = rem / Cin, c = rem % Cin;
    out[i] = (typename HT::T)w[((long)n * Cin + c) * k + j];
  }
}
const char* launch_pack_conv(const float* w, int N, int Cin, int k, void* out_h, int dtype, hipStream_t s) {
  const int blocks = (int)min((long)4096, ((long)N * Cin * k + 255) / 256);
  AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL(pack_conv_kernel<HT>, dim3(blocks), dim3(256), 0, s, w, N, Cin, k,
                                            (HT::T*)out_h));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// Positional conv: weight_v [C][cpg][k] (+ weight_g [k], weight-norm over dims (0,1))
// -> [C][j*cpg + c] with the per-tap scale g[j] / ||v[:,:,j]|| folded in.
__global__ void posconv_norm_kernel(const float* v, int C, int cpg, int k, float* norm) {
  const int j = blockIdx.x;
  float s = 0.f;
  for (long i = threadIdx.x; i < (long)C * cpg; i += blockDim.x) {
    const float x = v[i * k + j];
    s = fmaf(x, x, s);
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm[j] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}
template <class HT>
__global__ void pack_posconv_kernel(const float* v, const float* g, const float* norm, int C, int cpg, int k,
                                    typename HT::T* out) {
  const long total = (long)C * cpg * k;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int o = (int)(i / ((long)cpg * k));
    const int rem = (int)(i % ((long)cpg * k));
    const int j = rem / cpg, c = rem % cpg;
    float x = v[((long)o * cpg + c) * k + j];
    if (g) x = x * g[j] / norm[j];
    out[i] = (typename HT::T)x;
  }
}
const char* launch_pack_posconv(const float* v, const float* g, int C, int cpg, int k, float* norm_tmp, void* out_h,
                                int dtype, hipStream_t s) {
  if (g) hipLaunchKernelGGL(posconv_norm_kernel, dim3(k), dim3(256), 0, s, v, C, cpg, k, norm_tmp);
  const int blocks = (int)min((long)4096, ((long)C * cpg * k + 255) / 256);
  AFX_DISPATCH_HT(dtype, hipLaunchKernelGGL(pack_posconv_kernel<HT>, dim3(blocks), dim3(256), 0, s, v, g, norm_tmp, C,
                                            cpg, k, (HT::T*)out_h));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---- split precision (DT_FP16X3) operand preparation --------------------------------------------------------------------
// One workgroup per weight row: the row is read whole (registers -> LDS) before anything is written, so its pair form
// (afx_kernels.h: hi / lo halves interleaved in groups of 32) can take the place of the fp32 row it was made from.
__global__ __launch_bounds__(256) void split_weight_rows_kernel(float* w, int K, float* row_scale) {
  extern __shared__ float srow[];
  float* row = w + (long)blockIdx.x * K;
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = row[k];
    srow[k] = v;
    mx = fmaxf(mx, fabsf(v));
  }
  __shared__ float red[4];
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // 2^e with mx * 2^e in [8192, 16384): hi parts are far from fp16's overflow, lo parts of typical entries far from its subnormals
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) {
    int ex;
    (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
    e = 14 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  const float sc = ldexpf(1.0f, e);
  __syncthreads();  // (every thread has read its part of the row: the pair form overwrites it)
  _Float16* out = (_Float16*)row;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = srow[k] * sc;
    const _Float16 h = (_Float16)v;
    const long o = s3_pair_index(k);
    out[o] = h;
    out[o + 32] = (_Float16)(v - (float)h);
  }
  if (threadIdx.x == 0) row_scale[blockIdx.x] = ldexpf(1.0f, -e);
}
const char* launch_split_weight_rows(void* w, int N, int K, float* row_scale, hipStream_t s) {
  if (K > 12288 || K % 32) return "split_weight_rows: rows of at most 12288 elements, whole 32-element groups";
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)split_weight_rows_kernel, K * 4 + 64); e != hipSuccess) return hipGetErrorString(e);
  hipLaunchKernelGGL(split_weight_rows_kernel, dim3(N), dim3(256), (size_t)K * 4, s, (float*)w, K, row_scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// fp32 values -> their pair form: a thread turns 8 consecutive values (inside one 32-element group) into 16 B of hi and 16 B of lo halves
__global__ __launch_bounds__(256) void split_pairs_kernel(const float* __restrict__ x, long n8, _Float16* __restrict__ out, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long o = i * 8;
    const f32x4 a = *(const f32x4*)(x + o), b = *(const f32x4*)(x + o + 4);
    f16x8 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float va = a[r] * scale, vb = b[r] * scale;
      h[r] = (_Float16)va;
      h[4 + r] = (_Float16)vb;
      l[r] = (_Float16)(va - (float)h[r]);
      l[4 + r] = (_Float16)(vb - (float)h[4 + r]);
    }
    _Float16* hp = out + s3_pair_index(o);
    *(f16x8*)hp = h;
    *(f16x8*)(hp + 32) = l;
  }
}
const char* launch_split_pairs(const float* x, long n, void* pairs, float scale, hipStream_t s) {
  if (n <= 0) return nullptr;
  if (((size_t)x & 15) || ((size_t)pairs & 15) || (n & 31)) return "split_pairs: whole groups of 32 elements, 16-byte aligned";
  const long n8 = n / 8;
  const int blocks = (int)min((long)8192, (n8 + 255) / 256);
  hipLaunchKernelGGL(split_pairs_kernel, dim3(blocks), dim3(256), 0, s, x, n8, (_Float16*)pairs, scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace afx
