// Conformer student head kernels, gfx950.  The dense
// products of the block run on the shared MFMA GEMM; what is here is the non-GEMM
// remainder, all fp32: token assembly, Shaw relative-position attention,
// GLU + depthwise conv + BatchNorm + Swish, and the 2-way classifier.
#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

// ---------------------------------------------------------------------------------
// models/conformer_baseline.py:59-62 + :23-24 -- BatchNorm2d(1) (eval: one scalar
// scale/shift), SELU, then the class token prepended to every utterance.
// ---------------------------------------------------------------------------------
__global__ void conf_tokens_kernel(const float* __restrict__ ll, const float* __restrict__ cls, float bn_scale,
                                   float bn_shift, int T, int E, float* __restrict__ out, int raw) {
  const int b = blockIdx.y;
  const long n = (long)(T + 1) * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / E), c = (int)(i % E);
    float v;
    if (row == 0)
      v = cls[c];
    else if (raw)
      v = ll[((long)b * T + row - 1) * E + c];
    else
      v = selu(fmaf(ll[((long)b * T + row - 1) * E + c], bn_scale, bn_shift));
    out[(long)b * n + i] = v;
  }
}
const char* launch_conf_tokens(const float* ll, const float* cls, float bn_scale, float bn_shift, int B, int T,
                               int E, float* out, hipStream_t s, bool raw) {
  hipLaunchKernelGGL(conf_tokens_kernel, dim3(16, B), dim3(256), 0, s, ll, cls, bn_scale, bn_shift, T, E, out, raw ? 1 : 0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// Shaw relative-position multi-head attention (lucidrains ConformerBlock.attn):
//   logits[i][j] = (q_i . k_j + q_i . E[clamp(i-j, +-P) + P]) * dh^-0.5
// One workgroup per (utterance, head); K, V and the (2N-1)-row window of E live in
// LDS as fp32 (115 KB at N=200, dh=36 -- sized for CDNA4's 160 KB).  Thread i owns
// query row i: q in registers, online softmax, K/V rows are LDS broadcasts and the E
// row (i-j+N-1) is a conflict-free ds_read_b128 (row stride 36 dwords).
// The (N,N,dh) relative tensor of the reference is never materialised.
// ---------------------------------------------------------------------------------
// REL = false drops the relative term: plain softmax(q k^T / sqrt dh) v -- the fp32
// "exact mode" stand-in for the matrix-core mhsa_kernel of the wav2vec2 trunk.
template <int DH, class HT, bool REL>
__global__ __launch_bounds__(256) void conf_attn_kernel(const float* __restrict__ q, long ldq,
                                                        const float* __restrict__ kv, long ldkv,
                                                        const float* __restrict__ rel, int max_pos, int N, int H,
                                                        int KB, typename HT::T* __restrict__ out, long ldo,
                                                        const int* __restrict__ lens, int len_add) {
  typedef typename HT::T Tt;
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  const int Nrow = N;  // ragged batch: utterance b owns Nrow rows in memory, lens[b] + len_add of them are tokens;
  if (lens) N = lens[blockIdx.y] + len_add;  // from here on N is ITS token count (keys beyond are zeros and masked)

  // Keys go through LDS KB at a time (KB = N, one pass, whenever the sequence fits: 4-s clips; longer clips
  // -- test_duration_sec is a free config value, data/test_set.py -- take blocks of 64 keys and 256-query
  // chunks on blockIdx.z).  The running max / sum walk the keys in the same order either way, so the
  // blocked form gives the same bits as the one-pass form.
  const int i0 = blockIdx.z * 256;
  const int nq = min(256, N - i0);
  float* Ks = sm_f;                 // [KB][DH]
  float* Vs = Ks + (long)KB * DH;   // [KB][DH]
  float* Es = Vs + (long)KB * DH;   // [nq + KB - 1][DH]: row (i - j) - (i0 - (j0 + KB - 1))
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int inner = H * DH;
  const bool live = tid < nq;
  const int i = i0 + (live ? tid : nq - 1);
  const float scale = 1.0f / sqrtf((float)DH);
  float qv[DH], o[DH];
  const float* qrow = q + ((long)b * Nrow + i) * ldq + h * DH;
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    const f32x4 t = *(const f32x4*)(qrow + d);
    qv[d] = t[0] * scale; qv[d + 1] = t[1] * scale; qv[d + 2] = t[2] * scale; qv[d + 3] = t[3] * scale;
    o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
  }
  float m = -1e30f, l = 0.f;
  for (int jb = 0; jb < N; jb += KB) {
    const int nk = min(KB, N - jb);
    if (jb) __syncthreads();
    for (int idx = tid; idx < nk * (DH / 4); idx += 256) {
      const int j = idx / (DH / 4), d4 = idx % (DH / 4);
      const float* row = kv + ((long)b * Nrow + jb + j) * ldkv + h * DH + d4 * 4;
      *(f32x4*)(Ks + j * DH + d4 * 4) = *(const f32x4*)row;
      *(f32x4*)(Vs + j * DH + d4 * 4) = *(const f32x4*)(row + inner);
    }
    const int ebase = i0 - (jb + KB - 1);  // distance i - j of window row 0
    if (REL) {
      for (int idx = tid; idx < (nq + KB - 1) * (DH / 4); idx += 256) {
        const int r = idx / (DH / 4), d4 = idx % (DH / 4);
        int dist = r + ebase;
        dist = dist < -max_pos ? -max_pos : (dist > max_pos ? max_pos : dist);
        *(f32x4*)(Es + r * DH + d4 * 4) = *(const f32x4*)(rel + (long)(dist + max_pos) * DH + d4 * 4);
      }
    }
    __syncthreads();
    // keys are walked 4 at a time: four independent score chains (ILP for the single wave a
    // SIMD hosts here), one running-max update and one rescale of o per group
    for (int j0 = 0; j0 < nk; j0 += 4) {
      float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u < nk ? j0 + u : nk - 1;
        const float* kr = Ks + j * DH;
        const float* er = Es + (i - (jb + j) - ebase) * DH;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
          const f32x4 kk = *(const f32x4*)(kr + d);
          const f32x4 ee = REL ? *(const f32x4*)(er + d) : f32x4{0.f, 0.f, 0.f, 0.f};
          s0 = fmaf(qv[d], kk[0] + ee[0], s0);
          s1 = fmaf(qv[d + 1], kk[1] + ee[1], s1);
          s0 = fmaf(qv[d + 2], kk[2] + ee[2], s0);
          s1 = fmaf(qv[d + 3], kk[3] + ee[3], s1);
        }
        sc[u] = j0 + u < nk ? s0 + s1 : -1e30f;
      }
      const float mn = fmaxf(fmaxf(m, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
      const float corr = __expf(m - mn);
      float pr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) pr[u] = __expf(sc[u] - mn);
      l = fmaf(l, corr, (pr[0] + pr[1]) + (pr[2] + pr[3]));
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] *= corr;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u < nk ? j0 + u : nk - 1;  // masked keys have pr == 0
        const float* vr = Vs + j * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
          const f32x4 vv = *(const f32x4*)(vr + d);
          o[d] = fmaf(pr[u], vv[0], o[d]);
          o[d + 1] = fmaf(pr[u], vv[1], o[d + 1]);
          o[d + 2] = fmaf(pr[u], vv[2], o[d + 2]);
          o[d + 3] = fmaf(pr[u], vv[3], o[d + 3]);
        }
      }
      m = mn;
    }
  }
  if (!live) return;
  const float rl = 1.0f / l;
  Tt* orow = out + ((long)b * Nrow + i) * ldo + h * DH;
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    typename HT::V4 v4;
    v4[0] = (Tt)(o[d] * rl); v4[1] = (Tt)(o[d + 1] * rl); v4[2] = (Tt)(o[d + 2] * rl); v4[3] = (Tt)(o[d + 3] * rl);
    *(typename HT::V4*)(orow + d) = v4;
  }
}

static int g_conf_attn_block = 0;  // test knob: force the blocked form with this many keys per block (multiple of 4)
void conf_attn_set_block(int v) { g_conf_attn_block = v > 0 ? (v + 3) & ~3 : 0; }

template <int DH, class HT, bool REL>
static hipError_t launch_conf_attn_t(const float* q, long ldq, const float* kv, long ldkv, const float* rel,
                                     int max_pos, int B, int N, int H, void* out, long ldo, hipStream_t s,
                                     const int* lens, int len_add) {
  auto lds_of = [&](int kb) { return (int)((2L * kb + (REL ? min(256, N) + kb - 1 : 0)) * DH * sizeof(float)); };
  int KB = N;  // one pass when a 256-query workgroup sees the whole sequence and it fits
  if (g_conf_attn_block && g_conf_attn_block < N) KB = g_conf_attn_block;
  else if (N > 256 || lds_of(N) > 160 * 1024) KB = 64;
  const int lds = lds_of(KB);
  static LdsLimit lim;  // sticky per device: raised only when a larger request arrives (graph-capture friendly)
  if (hipError_t e = lim.ensure((const void*)conf_attn_kernel<DH, HT, REL>, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((conf_attn_kernel<DH, HT, REL>), dim3(H, B, (N + 255) / 256), dim3(256), lds, s, q, ldq, kv, ldkv, rel,
                     max_pos, N, H, KB, (typename HT::T*)out, ldo, lens, len_add);
  return hipGetLastError();
}

const char* launch_conf_attn(const float* q, long ldq, const float* kv, long ldkv, const float* rel, int max_pos,
                             int B, int N, int H, int dh, void* out_h, long ldo, int dtype, hipStream_t s,
                             const int* lens, int len_add) {
  if (N <= 0 || B <= 0 || B > 65535) return "conf_attn: bad shape";
  if ((ldq % 4) || (ldkv % 4) || (ldo % 4)) return "conf_attn: row strides must be multiples of 4";
  hipError_t e = hipSuccess;
#define AFX_CA(DHv)                                                                                             \
  AFX_DISPATCH_HT(dtype, e = launch_conf_attn_t<DHv, HT, true>(q, ldq, kv, ldkv, rel, max_pos, B, N, H, out_h, ldo, s, lens, len_add))
  if (!rel) {  // plain attention (exact-mode trunk): fp32 in, fp32 out, head dim 64
    if (dh != 64 || dtype != DT_FP32) return "conf_attn: the no-relative-term form is the fp32 trunk attention (dh 64)";
    e = launch_conf_attn_t<64, F32T, false>(q, ldq, kv, ldkv, nullptr, 0, B, N, H, out_h, ldo, s, lens, len_add);
  }
  else if (dh == 36) { AFX_CA(36); }
  else if (dh == 32) { AFX_CA(32); }
  else if (dh == 64) { AFX_CA(64); }
  else return "conf_attn: supported head dims are 32, 36 (emb 144 / 4 heads) and 64";
#undef AFX_CA
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// The same attention on the matrix cores (operand type fp16 / bf16, fp32 accumulation), for
// N <= 208 tokens and head dim <= 64.  One workgroup per (utterance, head); K (row-major,
// XOR-swizzled 128-B rows, head dim zero-padded to 64) and V^T live in LDS as in mhsa_kernel;
// a wave walks 16-query tiles:
//   S1^T = K Q^T                       13 key tiles x 2 k-steps
//   R^T  = E_win Q^T                   the relative term for ALL offsets the tile can see:
//                                      rows r = i - j + N - 1 in [i0, i0 + N + 14], fragments of
//                                      the fp16 embedding table straight from L2
//   skew                               R goes through a per-wave LDS tile [16][228] fp32 and comes
//                                      back as S2[i][j] = R[i][i - j + N - 1] (Transformer-XL's
//                                      relative shift), conflict-free (row stride 229 words is odd)
//   softmax, O^T = V^T P^T             as in mhsa_kernel; the lane keeps its query row throughout
// The reference's (N, N, dh) relative tensor is never formed; 75 MFMAs replace ~24 k fp32 FMAs
// per query tile.
// ---------------------------------------------------------------------------------
constexpr int CA_KEYS = 224, CA_VT_STRIDE = 232, CA_RS = 228, CA_NKT = 14, CA_KS = 7;

// NW waves per workgroup (4 or 7): the 13 query tiles of a 200-token head are dealt round-robin, 2 per wave with seven
// waves instead of 4 with four; 152 KB of LDS (a 14.6-KB skew tile per wave) -- one workgroup per CU either way, and at
// batch 64 there are exactly B x H = 256 of them.
template <class HT, int DH, int NW>
__global__ __launch_bounds__(64 * NW) void conf_attn_mfma_kernel(const float* __restrict__ q, long ldq,
                                                             const float* __restrict__ kv, long ldkv,
                                                             const typename HT::T* __restrict__ rel_h, int max_pos,
                                                             int N, int H, typename HT::T* __restrict__ out, long ldo,
                                                             const int* __restrict__ lens, int len_add) {
  typedef typename HT::T Tt;
  typedef typename HT::V8 V8;
  typedef typename HT::V4 V4;
  const int Nrow = N;  // ragged batch: utterance b owns Nrow rows in memory, lens[b] + len_add of them are tokens;
  if (lens) N = lens[blockIdx.y] + len_add;  // from here on N is ITS token count (keys beyond are zeros and masked)
  constexpr int DT = (DH + 15) / 16;  // 16-wide head-dim tiles of the output
  static_assert(DH % 4 == 0 && DH <= 64, "head dim");
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  char* sm = (char*)sm_f;
  char* k_lds = sm;                                        // [224][128 B]
  Tt* vt_lds = (Tt*)(sm + CA_KEYS * 128);                  // [16 DT][232]
  float* r_lds = (float*)(sm + CA_KEYS * 128 + 16 * DT * CA_VT_STRIDE * 2);  // [NW waves][16][228]
  constexpr int NT = 64 * NW;

  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = H * DH;
  // ---- stage K (fp16, swizzled, zero padded) and V^T ---------------------------------
  for (int i = tid; i < (CA_KEYS * 128 + 16 * DT * CA_VT_STRIDE * 2) / 16; i += NT) ((u32x4*)sm)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  {  // all of the thread's global loads first, then the LDS writes (one dependent L2 round trip, not eight)
    constexpr int IT = (209 * (DH / 4) + NT - 1) / NT;
    f32x4 kreg[IT], vreg[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      if (idx < N * (DH / 4)) {
        const int key = idx / (DH / 4), q4 = idx % (DH / 4);
        const float* row = kv + ((long)b * Nrow + key) * ldkv + h * DH + q4 * 4;
        kreg[it] = *(const f32x4*)row;
        vreg[it] = *(const f32x4*)(row + inner);
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      if (idx < N * (DH / 4)) {
        const int key = idx / (DH / 4), q4 = idx % (DH / 4);
        V4 kh;
#pragma unroll
        for (int i = 0; i < 4; ++i) kh[i] = (Tt)kreg[it][i];
        const int c = q4 >> 1;
        *(V4*)(k_lds + key * 128 + ((c ^ ((key >> 1) & 7)) * 16) + (q4 & 1) * 8) = kh;
#pragma unroll
        for (int i = 0; i < 4; ++i) vt_lds[(q4 * 4 + i) * CA_VT_STRIDE + key] = (Tt)vreg[it][i];
      }
    }
  }
  __syncthreads();

  const int ql = lane & 15, g = lane >> 4;
  const float scale = 1.0f / sqrtf((float)DH);
  float* rw = r_lds + wave * 16 * CA_RS;
  const int nqt = (N + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += NW) {
    const int q0 = qt * 16;
    int qrow = q0 + ql;
    qrow = qrow < N ? qrow : N - 1;
    // Q fragments (scaled, operand type): k-slot (g, j) of step ks <-> head dim 32 ks + 8 g + j
    V8 qf[2];
    {
      const float* qp = q + ((long)b * Nrow + qrow) * ldq + h * DH;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int d = ks * 32 + g * 8 + half * 4;
          f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
          if (d < DH) t = *(const f32x4*)(qp + d);
#pragma unroll
          for (int i = 0; i < 4; ++i) qf[ks][half * 4 + i] = (Tt)(t[i] * scale);
        }
    }
    // relative term for every offset this tile can see: r_local = r - q0, E row = r - (N-1) + max_pos
    // (round 4, late: the table rows of the NEXT pair of offset tiles -- L2 round trips -- are requested before this pair's MFMAs)
    auto read_e = [&](int rp, V8 (&ef)[2][2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int er = q0 + (rp + u) * 16 + ql - (N - 1);
        er = er < -max_pos ? -max_pos : (er > max_pos ? max_pos : er);
        const Tt* ep = rel_h + (long)(er + max_pos) * 64 + g * 8;
        ef[u][0] = *(const V8*)ep;
        ef[u][1] = *(const V8*)(ep + 32);
      }
    };
    V8 efa[2][2], efb[2][2];
    read_e(0, efa);
#pragma unroll
    for (int rp = 0; rp < CA_NKT; rp += 2) {
      V8(&ef)[2][2] = (rp & 2) ? efb : efa;
      V8(&efn)[2][2] = (rp & 2) ? efa : efb;
      if (rp + 2 < CA_NKT) read_e(rp + 2, efn);
      f32x4 r2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) r2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) r2[u] = HT::mfma(ef[u][ks], qf[ks], r2[u]);
#pragma unroll
      for (int u = 0; u < 2; ++u) *(f32x4*)(rw + ql * CA_RS + (rp + u) * 16 + g * 4) = r2[u];
    }
    // S^T tiles: s[kt][c] = S1[q0+ql][16kt + 4g + c], two key tiles at a time
    f32x4 s[CA_NKT];
    auto read_k = [&](int kp, V8 (&kf)[2][2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int krow = (kp + u) * 16 + ql;
        const int sw = (krow >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[u][ks] = *(const V8*)(k_lds + krow * 128 + (((ks * 4 + g) ^ sw) * 16));
      }
    };
    V8 kfa[2][2], kfb[2][2];
    read_k(0, kfa);
#pragma unroll
    for (int kp = 0; kp < CA_NKT; kp += 2) {
      V8(&kf)[2][2] = (kp & 2) ? kfb : kfa;
      V8(&kfn)[2][2] = (kp & 2) ? kfa : kfb;
      if (kp + 2 < CA_NKT) read_k(kp + 2, kfn);  // the next pair's fragments under this pair's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) s[kp + u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) s[kp + u] = HT::mfma(kf[u][ks], qf[ks], s[kp + u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // add the shifted relative term, mask, softmax (the wave's own R tile: no barrier needed,
    // LDS accesses of one wave complete in order)
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < CA_NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int key = kt * 16 + g * 4 + c;
        int rl = ql + (N - 1) - key;
        rl = rl < 0 ? 0 : rl;
        const float v = key < N ? s[kt][c] + rw[ql * CA_RS + rl] : -1e30f;
        s[kt][c] = v;
        mx = fmaxf(mx, v);
      }
    mx = rows_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < CA_NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e = __expf(s[kt][c] - mx);
        s[kt][c] = e;
        sum += e;
      }
    sum = rows_sum(sum);
    const float rinv = 1.0f / sum;
    // O = P V : k-slot (g, jj) of step s2 <-> key 32*s2 + 16*(jj>>2) + 4g + (jj&3)
    f32x4 o[DT];
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto read_v = [&](int s2, V8 (&vf)[DT]) {
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) {
        const Tt* vr = vt_lds + (nt * 16 + ql) * CA_VT_STRIDE + s2 * 32 + g * 4;
        const V4 lo = *(const V4*)vr;
        const V4 hi = *(const V4*)(vr + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          vf[nt][c] = lo[c];
          vf[nt][4 + c] = hi[c];
        }
      }
    };
    V8 vfa[DT], vfb[DT];
    read_v(0, vfa);
#pragma unroll
    for (int s2 = 0; s2 < CA_KS; ++s2) {
      V8(&vf)[DT] = (s2 & 1) ? vfb : vfa;
      V8(&vfn)[DT] = (s2 & 1) ? vfa : vfb;
      if (s2 + 1 < CA_KS) read_v(s2 + 1, vfn);  // the next key step's fragments under this step's MFMAs
      V8 pf;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pf[c] = (Tt)s[2 * s2][c];
        pf[4 + c] = (Tt)s[2 * s2 + 1][c];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) o[nt] = HT::mfma(vf[nt], pf, o[nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int qi = q0 + ql;
    if (qi < N) {
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) {
        const int d = nt * 16 + g * 4;
        if (d < DH) {
          V4 hv;
#pragma unroll
          for (int c = 0; c < 4; ++c) hv[c] = (Tt)(o[nt][c] * rinv);
          *(V4*)(out + ((long)b * Nrow + qi) * ldo + h * DH + d) = hv;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// The same one-pass Shaw attention in SPLIT PRECISION (the engine's dtype "fp16x3", round 4): fp32 q | k | v rows and an fp32 table
// (rows padded to 64) in, fp32 rows out; every product -- E Q^T, K Q^T, V^T P^T -- as three fp16 MFMAs on hi / lo halves
// (x ~ xh + xl: xl.yh + xh.yl + xh.yh, the small terms first), K and V^T staged as hi and lo images, Q / E / P split in registers.
// It replaces the fp32 VALU kernel in that mode for N <= 209 (147 -> see DESIGN section 4, us per launch at B = 64).  LDS: two K
// images, two V^T images and the waves' relative-term tiles = 156.5 KB with FOUR waves (seven do not fit).
// ---------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void conf_attn_split_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ kv, long ldkv,
                                                                  const float* __restrict__ rel64, int max_pos, int N, int H,
                                                                  float* __restrict__ out, long ldo, const int* __restrict__ lens, int len_add) {
  typedef _Float16 Tt;
  typedef f16x8 V8;
  typedef f16x4 V4;
  const int Nrow = N;
  if (lens) N = lens[blockIdx.y] + len_add;
  constexpr int DT = (DH + 15) / 16;
  static_assert(DH % 4 == 0 && DH <= 64, "head dim");
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  char* sm = (char*)sm_f;
  constexpr int KB = CA_KEYS * 128, VB = 16 * DT * CA_VT_STRIDE * 2;
  char* k_lds[2] = {sm, sm + KB};                                        // [hi / lo][224][128 B]
  Tt* vt_lds[2] = {(Tt*)(sm + 2 * KB), (Tt*)(sm + 2 * KB + VB)};         // [hi / lo][16 DT][232]
  float* r_lds = (float*)(sm + 2 * KB + 2 * VB);                         // [NW waves][16][228]
  constexpr int NT = 64 * NW;
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = H * DH;
  auto split4 = [](const f32x4& a, V4& hi, V4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hi[i] = (Tt)a[i];
      lo[i] = (Tt)(a[i] - (float)hi[i]);
    }
  };
  auto split8 = [](const f32x4& a, const f32x4& c, V8& hi, V8& lo) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hi[r] = (Tt)a[r];
      hi[4 + r] = (Tt)c[r];
      lo[r] = (Tt)(a[r] - (float)hi[r]);
      lo[4 + r] = (Tt)(c[r] - (float)hi[4 + r]);
    }
  };
  // ---- stage K (hi / lo, swizzled, zero padded) and V^T (hi / lo) -----------------------
  for (int i = tid; i < (2 * KB + 2 * VB) / 16; i += NT) ((u32x4*)sm)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  {
    constexpr int IT = (209 * (DH / 4) + NT - 1) / NT;
    f32x4 kreg[IT], vreg[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      if (idx < N * (DH / 4)) {
        const int key = idx / (DH / 4), q4 = idx % (DH / 4);
        const float* row = kv + ((long)b * Nrow + key) * ldkv + h * DH + q4 * 4;
        kreg[it] = *(const f32x4*)row;
        vreg[it] = *(const f32x4*)(row + inner);
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      if (idx < N * (DH / 4)) {
        const int key = idx / (DH / 4), q4 = idx % (DH / 4);
        V4 kh, kl, vh, vl;
        split4(kreg[it], kh, kl);
        split4(vreg[it], vh, vl);
        const int c = q4 >> 1;
        const int off = key * 128 + ((c ^ ((key >> 1) & 7)) * 16) + (q4 & 1) * 8;
        *(V4*)(k_lds[0] + off) = kh;
        *(V4*)(k_lds[1] + off) = kl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vt_lds[0][(q4 * 4 + i) * CA_VT_STRIDE + key] = vh[i];
          vt_lds[1][(q4 * 4 + i) * CA_VT_STRIDE + key] = vl[i];
        }
      }
    }
  }
  __syncthreads();

  const int ql = lane & 15, g = lane >> 4;
  const float scale = 1.0f / sqrtf((float)DH);
  float* rw = r_lds + wave * 16 * CA_RS;
  const int nqt = (N + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += NW) {
    const int q0 = qt * 16;
    int qrow = q0 + ql;
    qrow = qrow < N ? qrow : N - 1;
    // Q fragments (hi / lo): k-slot (g, j) of step ks <-> head dim 32 ks + 8 g + j
    V8 qh[2], qlo[2];
    {
      const float* qp = q + ((long)b * Nrow + qrow) * ldq + h * DH;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 t[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int d = ks * 32 + g * 8 + half * 4;
          t[half] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (d < DH) t[half] = *(const f32x4*)(qp + d);  // (unscaled: 1 / sqrt(dh) is applied to the fp32 products below)
        }
        split8(t[0], t[1], qh[ks], qlo[ks]);
      }
    }
    // relative term for every offset this tile can see (table rows from L2, fp32, split here); the next pair's rows are requested
    // before this pair's MFMAs
    auto read_e = [&](int rp, f32x4 (&ef)[2][2][2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int er = q0 + (rp + u) * 16 + ql - (N - 1);
        er = er < -max_pos ? -max_pos : (er > max_pos ? max_pos : er);
        const float* ep = rel64 + (long)(er + max_pos) * 64 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          ef[u][ks][0] = *(const f32x4*)(ep + ks * 32);
          ef[u][ks][1] = *(const f32x4*)(ep + ks * 32 + 4);
        }
      }
    };
    f32x4 efa[2][2][2], efb[2][2][2];
    read_e(0, efa);
#pragma unroll
    for (int rp = 0; rp < CA_NKT; rp += 2) {
      f32x4(&ef)[2][2][2] = (rp & 2) ? efb : efa;
      f32x4(&efn)[2][2][2] = (rp & 2) ? efa : efb;
      if (rp + 2 < CA_NKT) read_e(rp + 2, efn);
      f32x4 r2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) r2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          V8 eh, el;
          split8(ef[u][ks][0], ef[u][ks][1], eh, el);
          r2[u] = FP16::mfma(el, qh[ks], r2[u]);
          r2[u] = FP16::mfma(eh, qlo[ks], r2[u]);
          r2[u] = FP16::mfma(eh, qh[ks], r2[u]);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) *(f32x4*)(rw + ql * CA_RS + (rp + u) * 16 + g * 4) = r2[u] * scale;
    }
    // S^T tiles: s[kt][c] = S1[q0+ql][16kt + 4g + c]; one key tile's four fragments ahead of the MFMAs
    f32x4 s[CA_NKT];
    auto read_k = [&](int kt, V8 (&kf)[2][2]) {
      const int krow = kt * 16 + ql, sw = (krow >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = krow * 128 + (((ks * 4 + g) ^ sw) * 16);
        kf[ks][0] = *(const V8*)(k_lds[0] + off);
        kf[ks][1] = *(const V8*)(k_lds[1] + off);
      }
    };
    V8 kfa[2][2], kfb[2][2];
    read_k(0, kfa);
#pragma unroll
    for (int kt = 0; kt < CA_NKT; ++kt) {
      V8(&kf)[2][2] = (kt & 1) ? kfb : kfa;
      V8(&kfn)[2][2] = (kt & 1) ? kfa : kfb;
      if (kt + 1 < CA_NKT) read_k(kt + 1, kfn);
      __builtin_amdgcn_sched_barrier(0);
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s[kt] = FP16::mfma(kf[ks][1], qh[ks], s[kt]);
        s[kt] = FP16::mfma(kf[ks][0], qlo[ks], s[kt]);
        s[kt] = FP16::mfma(kf[ks][0], qh[ks], s[kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // add the shifted relative term, mask, softmax (the wave's own R tile: LDS accesses of one wave complete in order)
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < CA_NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int key = kt * 16 + g * 4 + c;
        int rl = ql + (N - 1) - key;
        rl = rl < 0 ? 0 : rl;
        const float v = key < N ? fmaf(s[kt][c], scale, rw[ql * CA_RS + rl]) : -1e30f;
        s[kt][c] = v;
        mx = fmaxf(mx, v);
      }
    mx = rows_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < CA_NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e = __builtin_amdgcn_exp2f((s[kt][c] - mx) * 1.4426950408889634f);
        s[kt][c] = e;
        sum += e;
      }
    sum = rows_sum(sum);
    const float rinv = 1.0f / sum;
    // O = P V : k-slot (g, jj) of step s2 <-> key 32*s2 + 16*(jj>>2) + 4g + (jj&3)
    f32x4 o[DT];
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto read_v = [&](int s2, V8 (&vf)[DT][2]) {
#pragma unroll
      for (int nt = 0; nt < DT; ++nt)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          const Tt* vr = vt_lds[hl] + (nt * 16 + ql) * CA_VT_STRIDE + s2 * 32 + g * 4;
          const V4 lo = *(const V4*)vr;
          const V4 hi = *(const V4*)(vr + 16);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            vf[nt][hl][c] = lo[c];
            vf[nt][hl][4 + c] = hi[c];
          }
        }
    };
    V8 vfa[DT][2], vfb[DT][2];
    read_v(0, vfa);
#pragma unroll
    for (int s2 = 0; s2 < CA_KS; ++s2) {
      V8(&vf)[DT][2] = (s2 & 1) ? vfb : vfa;
      V8(&vfn)[DT][2] = (s2 & 1) ? vfa : vfb;
      if (s2 + 1 < CA_KS) read_v(s2 + 1, vfn);
      V8 ph, pl;
      split8(s[2 * s2], s[2 * s2 + 1], ph, pl);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) {
        o[nt] = FP16::mfma(vf[nt][1], ph, o[nt]);
        o[nt] = FP16::mfma(vf[nt][0], pl, o[nt]);
        o[nt] = FP16::mfma(vf[nt][0], ph, o[nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int qi = q0 + ql;
    if (qi < N) {
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) {
        const int d = nt * 16 + g * 4;
        if (d < DH) *(f32x4*)(out + ((long)b * Nrow + qi) * ldo + h * DH + d) = o[nt] * rinv;
      }
#ifdef SHAW_DBG  // diagnostics: the row's maximum logit and softmax denominator in head dims 0 / 1 (tools/diag_shaw_split_err.py)
      if (g == 0) {
        out[((long)b * Nrow + qi) * ldo + h * DH + 0] = mx;
        out[((long)b * Nrow + qi) * ldo + h * DH + 1] = sum;
      }
#endif
    }
  }
}
// fp32 rows in / out, the table as fp32 rows padded to 64 (launch_pack_linear(..., 64, ..., DT_FP32)); N <= 209, head dim 36
const char* launch_conf_attn_split(const float* q, long ldq, const float* kv, long ldkv, const float* rel64, int max_pos, int B, int N,
                                   int H, int dh, float* out, long ldo, hipStream_t s, const int* lens, int len_add) {
  if (dh != 36) return "conf_attn_split: built for head dim 36 (emb 144 / 4 heads)";
  if (N <= 0 || N + 15 > CA_KEYS || B <= 0 || B > 65535) return "conf_attn_split: 1..209 tokens";
  if ((ldq % 4) || (ldkv % 4) || (ldo % 4)) return "conf_attn_split: row strides must be multiples of 4";
  constexpr int DT = 3, NW = 4;
  const int lds = 2 * CA_KEYS * 128 + 2 * 16 * DT * CA_VT_STRIDE * 2 + NW * 16 * CA_RS * 4;
  static LdsLimit lim;
  hipError_t e = lim.ensure((const void*)conf_attn_split_kernel<36, NW>, lds);
  if (e == hipSuccess) {
    hipLaunchKernelGGL((conf_attn_split_kernel<36, NW>), dim3(H, B), dim3(64 * NW), lds, s, q, ldq, kv, ldkv, rel64, max_pos, N, H, out, ldo, lens, len_add);
    e = hipGetLastError();
  }
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// The matrix-core Shaw attention for sequences of any length (clips beyond 4 s): one workgroup per
// (utterance, head, 64-query chunk), a wave keeps ONE 16-query tile -- Q fragments, output accumulators and
// row statistics in registers -- while the keys pass through LDS 128 at a time.  Per (query tile, key
// block) the relative term is needed for the 143 distances i - j in [q0 - j0 - 127, q0 - j0 + 15]:
// nine 16-row tiles of E (clamped to +-max_pos, fragments from L2) times Q^T, skewed through the wave's
// own LDS tile exactly as in the one-pass kernel; then the running maximum / sum update of mhsa_long_kernel.
// LDS 66 KB: two workgroups per CU.
// ---------------------------------------------------------------------------------
constexpr int CAL_KB = 128, CAL_VT_STRIDE = 136, CAL_RS = 148, CAL_NKT = 8, CAL_KS = 4, CAL_RT = 9;

template <class HT, int DH>
__global__ __launch_bounds__(256, 2) void conf_attn_mfma_long_kernel(const float* __restrict__ q, long ldq,
                                                                  const float* __restrict__ kv, long ldkv,
                                                                  const typename HT::T* __restrict__ rel_h, int max_pos,
                                                                  int N, int H, typename HT::T* __restrict__ out, long ldo,
                                                                  const int* __restrict__ lens, int len_add) {
  typedef typename HT::T Tt;
  typedef typename HT::V8 V8;
  typedef typename HT::V4 V4;
  const int Nrow = N;  // ragged batch: utterance b owns Nrow rows in memory, lens[b] + len_add of them are tokens;
  if (lens) N = lens[blockIdx.y] + len_add;  // from here on N is ITS token count (keys beyond are zeros and masked)
  constexpr int DT = (DH + 15) / 16;
  static_assert(DH % 4 == 0 && DH <= 64, "head dim");
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  char* sm = (char*)sm_f;
  char* k_lds = sm;                                                           // [128][128 B]
  Tt* vt_lds = (Tt*)(sm + CAL_KB * 128);                                      // [16 DT][136]
  float* r_lds = (float*)(sm + CAL_KB * 128 + 16 * DT * CAL_VT_STRIDE * 2);   // [4 waves][16][148]
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = H * DH;
  const int ql = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.z * 64 + wave * 16;
  const float scale = 1.0f / sqrtf((float)DH);
  float* rw = r_lds + wave * 16 * CAL_RS;
  // head dims DH..63 of K and rows DH..16 DT of V^T are never written again: zero once
  for (int i = tid; i < (CAL_KB * 128 + 16 * DT * CAL_VT_STRIDE * 2) / 16; i += 256) ((u32x4*)sm)[i] = u32x4{0u, 0u, 0u, 0u};
  V8 qf[2];
  {
    const int qrow = q0 + ql < N ? q0 + ql : N - 1;
    const float* qp = q + ((long)b * Nrow + qrow) * ldq + h * DH;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int d = ks * 32 + g * 8 + half * 4;
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d < DH) t = *(const f32x4*)(qp + d);
#pragma unroll
        for (int i = 0; i < 4; ++i) qf[ks][half * 4 + i] = (Tt)(t[i] * scale);
      }
  }
  float m = -1e30f, l = 0.f;
  f32x4 o[DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j0 = 0; j0 < N; j0 += CAL_KB) {
    __syncthreads();  // the zero fill (first block) / every wave done with the previous block
    {
      const int nk = min(CAL_KB, N - j0);
      constexpr int IT = (CAL_KB * (DH / 4) + 255) / 256;
      f32x4 kreg[IT], vreg[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * 256;
        if (idx < nk * (DH / 4)) {
          const int key = idx / (DH / 4), q4 = idx % (DH / 4);
          const float* row = kv + ((long)b * Nrow + j0 + key) * ldkv + h * DH + q4 * 4;
          kreg[it] = *(const f32x4*)row;
          vreg[it] = *(const f32x4*)(row + inner);
        }
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * 256;
        if (idx < nk * (DH / 4)) {
          const int key = idx / (DH / 4), q4 = idx % (DH / 4);
          V4 kh;
#pragma unroll
          for (int i = 0; i < 4; ++i) kh[i] = (Tt)kreg[it][i];
          const int c = q4 >> 1;
          *(V4*)(k_lds + key * 128 + ((c ^ ((key >> 1) & 7)) * 16) + (q4 & 1) * 8) = kh;
#pragma unroll
          for (int i = 0; i < 4; ++i) vt_lds[(q4 * 4 + i) * CAL_VT_STRIDE + key] = (Tt)vreg[it][i];
        }
      }
    }
    __syncthreads();
    if (q0 >= N) continue;  // wave-uniform: a tile past the end only helps staging (keys past N keep finite stale values: p = 0)

    // relative term: window row r <-> distance base + r
    const int base = q0 - j0 - (CAL_KB - 1);
#pragma unroll
    for (int rp = 0; rp < CAL_RT; ++rp) {
      int er = base + rp * 16 + ql;
      er = er < -max_pos ? -max_pos : (er > max_pos ? max_pos : er);
      const Tt* ep = rel_h + (long)(er + max_pos) * 64 + g * 8;
      const V8 e0 = *(const V8*)ep, e1 = *(const V8*)(ep + 32);
      f32x4 r2 = f32x4{0.f, 0.f, 0.f, 0.f};
      r2 = HT::mfma(e0, qf[0], r2);
      r2 = HT::mfma(e1, qf[1], r2);
      *(f32x4*)(rw + ql * CAL_RS + rp * 16 + g * 4) = r2;
    }
    f32x4 s[CAL_NKT];
#pragma unroll
    for (int kp = 0; kp < CAL_NKT; kp += 2) {
      V8 kf[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int krow = (kp + u) * 16 + ql;
        const int sw = (krow >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[u][ks] = *(const V8*)(k_lds + krow * 128 + (((ks * 4 + g) ^ sw) * 16));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) s[kp + u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) s[kp + u] = HT::mfma(kf[u][ks], qf[ks], s[kp + u]);
    }
    float bm = -1e30f;
#pragma unroll
    for (int kt = 0; kt < CAL_NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kl = kt * 16 + g * 4 + c;
        const float v = j0 + kl < N ? s[kt][c] + rw[ql * CAL_RS + ql + (CAL_KB - 1) - kl] : -1e30f;
        s[kt][c] = v;
        bm = fmaxf(bm, v);
      }
    const float mn = fmaxf(m, rows_max(bm));
    const float corr = __expf(m - mn);
    float bl = 0.f;
#pragma unroll
    for (int kt = 0; kt < CAL_NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e = __expf(s[kt][c] - mn);
        s[kt][c] = e;
        bl += e;
      }
    l = fmaf(l, corr, rows_sum(bl));
    m = mn;
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) o[nt] *= corr;
#pragma unroll
    for (int s2 = 0; s2 < CAL_KS; ++s2) {
      V8 pf;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pf[c] = (Tt)s[2 * s2][c];
        pf[4 + c] = (Tt)s[2 * s2 + 1][c];
      }
      V8 vf[DT];
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) {
        const Tt* vr = vt_lds + (nt * 16 + ql) * CAL_VT_STRIDE + s2 * 32 + g * 4;
        const V4 lo = *(const V4*)vr;
        const V4 hi = *(const V4*)(vr + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          vf[nt][c] = lo[c];
          vf[nt][4 + c] = hi[c];
        }
      }
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) o[nt] = HT::mfma(vf[nt], pf, o[nt]);
    }
  }
  const int qi = q0 + ql;
  if (qi < N) {
    const float rinv = 1.0f / l;
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) {
      const int d = nt * 16 + g * 4;
      if (d < DH) {
        V4 hv;
#pragma unroll
        for (int c = 0; c < 4; ++c) hv[c] = (Tt)(o[nt][c] * rinv);
        *(V4*)(out + ((long)b * Nrow + qi) * ldo + h * DH + d) = hv;
      }
    }
  }
}

static int g_conf_attn_force_long = 0;  // test knob: the blocked kernel at every length
void conf_attn_mfma_set_force_long(int v) { g_conf_attn_force_long = v != 0; }
static int g_conf_attn_waves = 7;  // A/B knob: waves per workgroup of the one-pass kernel (4 or 7)
void conf_attn_mfma_set_waves(int v) { g_conf_attn_waves = v == 4 ? 4 : 7; }

const char* launch_conf_attn_mfma(const float* q, long ldq, const float* kv, long ldkv, const void* rel_h, int max_pos,
                                  int B, int N, int H, int dh, void* out_h, long ldo, int dtype, hipStream_t s,
                                  const int* lens, int len_add) {
  if (dh != 36) return "conf_attn_mfma: built for head dim 36 (emb 144 / 4 heads)";
  if (N <= 0 || B <= 0 || B > 65535) return "conf_attn_mfma: bad shape";
  if (dtype == DT_FP32) return "conf_attn_mfma: half-precision operands only";
  if ((ldq % 4) || (ldkv % 4) || (ldo % 4)) return "conf_attn_mfma: row strides must be multiples of 4";
  constexpr int DT = 3;
  if (N + 15 > CA_KEYS || g_conf_attn_force_long) {  // beyond 209 tokens: keys in blocks of 128
    const int ldsl = CAL_KB * 128 + 16 * DT * CAL_VT_STRIDE * 2 + 4 * 16 * CAL_RS * 4;
    hipError_t el = hipSuccess;
    static LdsLimit liml[2];
    dim3 grid(H, B, (N + 63) / 64);
    if (dtype == DT_BF16) {
      el = liml[0].ensure((const void*)conf_attn_mfma_long_kernel<BF16, 36>, ldsl);
      if (el == hipSuccess)
        hipLaunchKernelGGL((conf_attn_mfma_long_kernel<BF16, 36>), grid, dim3(256), ldsl, s, q, ldq, kv, ldkv, (const __bf16*)rel_h,
                           max_pos, N, H, (__bf16*)out_h, ldo, lens, len_add);
    } else {
      el = liml[1].ensure((const void*)conf_attn_mfma_long_kernel<FP16, 36>, ldsl);
      if (el == hipSuccess)
        hipLaunchKernelGGL((conf_attn_mfma_long_kernel<FP16, 36>), grid, dim3(256), ldsl, s, q, ldq, kv, ldkv, (const _Float16*)rel_h,
                           max_pos, N, H, (_Float16*)out_h, ldo, lens, len_add);
    }
    if (el == hipSuccess) el = hipGetLastError();
    return el == hipSuccess ? nullptr : hipGetErrorString(el);
  }
  const int nw = N > 64 && g_conf_attn_waves != 4 ? 7 : 4;  // more than 4 query tiles: seven waves
  const int lds = CA_KEYS * 128 + 16 * DT * CA_VT_STRIDE * 2 + nw * 16 * CA_RS * 4;
  hipError_t e = hipSuccess;
  static LdsLimit lim[4];
#define AFX_CAM(HTv, NWv, slot, Tv)                                                                                      \
  do {                                                                                                                   \
    e = lim[slot].ensure((const void*)conf_attn_mfma_kernel<HTv, 36, NWv>, lds);                                         \
    if (e == hipSuccess)                                                                                                 \
      hipLaunchKernelGGL((conf_attn_mfma_kernel<HTv, 36, NWv>), dim3(H, B), dim3(64 * NWv), lds, s, q, ldq, kv, ldkv,    \
                         (const Tv*)rel_h, max_pos, N, H, (Tv*)out_h, ldo, lens, len_add);                               \
  } while (0)
  if (dtype == DT_BF16) {
    if (nw == 7) AFX_CAM(BF16, 7, 0, __bf16); else AFX_CAM(BF16, 4, 1, __bf16);
  } else {
    if (nw == 7) AFX_CAM(FP16, 7, 2, _Float16); else AFX_CAM(FP16, 4, 3, _Float16);
  }
#undef AFX_CAM
  if (e == hipSuccess) e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// ConformerConvModule middle: GLU(channel) -> depthwise Conv1d(k, "same" pad
// (k/2, k/2 - (k+1)%2)) -> BatchNorm1d (eval, folded to scale/shift) -> Swish.
// One workgroup per (utterance, 32-channel slab): the gated activations for every
// frame of the slab sit in LDS with the zero padding materialised.
// ---------------------------------------------------------------------------------
template <class HT>
__global__ __launch_bounds__(256) void conf_dwconv_kernel(const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ bn_scale,
                                                          const float* __restrict__ bn_shift, int N, int NC, int C, int k,
                                                          typename HT::T* __restrict__ out, long ldo,
                                                          const int* __restrict__ lens, int len_add) {
  typedef typename HT::T Tt;
  const int Nrow = N;  // ragged batch: utterance b owns Nrow rows in memory, lens[b] + len_add of them are tokens;
  if (lens) N = lens[blockIdx.y] + len_add;  // from here on N is ITS token count (keys beyond are zeros and masked)
  constexpr bool EXACT = sizeof(Tt) == 4;  // fp32 "exact mode": accurate exp / division
  constexpr int KMAX = 31, TB = 8;         // taps held in registers; outputs per register window
  extern __shared__ __attribute__((aligned(16))) float sm_f[];
  float* sm = sm_f;
  const int c0 = blockIdx.x * 32, b = blockIdx.y, tid = threadIdx.x;
  const int pl = k / 2;
  // blockIdx.z walks the sequence NC frames at a time (NC = N up to 1024 frames; longer clips take several
  // chunks, each staging its own k - 1 halo rows)
  const int f0 = blockIdx.z * NC, nc = min(NC, N - f0);
  const int rows = nc + KMAX - 1 + TB;  // zero rows behind the chunk let the last window read freely
  float* us = sm;                       // [rows][32]: gated activations, row r <-> frame f0 + r - pl
  const int cl = tid & 31, tl = tid >> 5;
  const int c = c0 + cl;
  const bool cok = c < C;
  for (int r = tl; r < rows; r += 8) {
    const int t = f0 + r - pl;
    float u = 0.f;
    if (cok && t >= 0 && t < N) {
      const float* row = x + ((long)b * Nrow + t) * ldx;
      u = row[c] * (EXACT ? sigmoid_acc(row[C + c]) : sigmoid_fast(row[C + c]));
    }
    us[r * 32 + cl] = u;
  }
  // this channel's taps in registers (zero beyond k), BatchNorm folded to scale/shift
  float wr[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) wr[j] = (cok && j < k) ? w[(long)c * k + j] : 0.f;
  const float bi = cok ? bias[c] : 0.f, sc = cok ? bn_scale[c] : 0.f, sh = cok ? bn_shift[c] : 0.f;
  __syncthreads();
  // each thread owns a contiguous run of frames and walks it TB outputs at a time: one window of
  // TB + 30 inputs comes from LDS, the TB x 31 FMAs run from registers (the per-tap LDS reads of
  // the first version, two per FMA, were what bounded this kernel)
  const int per = (nc + 7) / 8;
  const int t_end = min(nc, (tl + 1) * per);
  for (int t0 = tl * per; t0 < t_end; t0 += TB) {
    float win[TB + KMAX - 1];
#pragma unroll
    for (int i = 0; i < TB + KMAX - 1; ++i) win[i] = us[(t0 + i) * 32 + cl];
    float acc[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) acc[u] = bi;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
#pragma unroll
      for (int u = 0; u < TB; ++u) acc[u] = fmaf(wr[j], win[u + j], acc[u]);
    if (cok) {
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const float y = fmaf(acc[u], sc, sh);
        if (t0 + u < t_end) out[((long)b * Nrow + f0 + t0 + u) * ldo + c] = (Tt)(EXACT ? swish(y) : swish_fast(y));
      }
    }
  }
}

const char* launch_conf_dwconv(const float* x, long ldx, const float* w, const float* bias, const float* bn_scale,
                               const float* bn_shift, int B, int N, int C, int k, void* out_h, long ldo, int dtype,
                               hipStream_t s, const int* lens, int len_add) {
  if (k > 31) return "conf_dwconv: depthwise kernels up to 31 taps";
  if (N <= 0 || B <= 0 || B > 65535) return "conf_dwconv: bad shape";
  const int NC = N < 1024 ? N : 1024;
  const int lds = (NC + 31 - 1 + 8) * 32 * (int)sizeof(float);
  dim3 grid((C + 31) / 32, B, (N + NC - 1) / NC);
  hipError_t e = hipSuccess;
  static LdsLimit lim[3];
  if (dtype < 0 || dtype > 2) return "conf_dwconv: unknown dtype";
  AFX_DISPATCH_HT(dtype, {
    e = lim[dtype].ensure((const void*)conf_dwconv_kernel<HT>, lds);
    if (e == hipSuccess)
      hipLaunchKernelGGL(conf_dwconv_kernel<HT>, grid, dim3(256), lds, s, x, ldx, w, bias, bn_scale, bn_shift, N, NC, C, k,
                         (HT::T*)out_h, ldo, lens, len_add);
  });
  if (e == hipSuccess) e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------
// Tiny fp32 linear (classifier heads): out[r][n] = b[n] + x[r*row_stride + :K] . w[n]
// One wave per output row; K is walked 64 lanes at a time.
// ---------------------------------------------------------------------------------
__global__ void small_linear_kernel(const float* __restrict__ x, long row_stride, int rows, int K,
                                    const float* __restrict__ w, const float* __restrict__ b, int N,
                                    float* __restrict__ out, int* __restrict__ nonfinite) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (long)r * row_stride;
  for (int n = 0; n < N; ++n) {
    float a = 0.f;
    for (int kk = lane; kk < K; kk += 64) a = fmaf(xr[kk], w[(long)n * K + kk], a);
    a = wave_sum(a);
    if (lane == 0) {
      a += b ? b[n] : 0.f;
      out[(long)r * N + n] = a;
      if (nonfinite && !(fabsf(a) <= 3.0e38f)) atomicAdd(nonfinite, 1);  // (NaN fails the compare too)
    }
  }
}
const char* launch_small_linear(const float* x, long row_stride, int rows, int K, const float* w, const float* b,
                                int N, float* out, hipStream_t s, int* nonfinite) {
  hipLaunchKernelGGL(small_linear_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, row_stride, rows, K, w, b, N,
                     out, nonfinite);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace afx
