// Conformer block, row-local parts fused, gfx950.
//
// Everything in a Conformer block except the attention itself and the depthwise conv acts on
// one token row at a time.  A wave owns 16 token rows and carries them through a whole chain
// of LayerNorm -> Linear -> activation -> Linear -> residual steps WITHOUT leaving its
// registers: the fp32 residual rows live in the MFMA accumulator layout (lane = row l&15,
// 4 consecutive columns 4(l>>4)..+3 of every 16-column tile); a LayerNorm is a 4-lane
// reduction (v_permlane16/32_swap); turning 32 columns of accumulator into the next product's
// A operand is one v_permlane16_swap per register (the wide-store exchange of the GEMM
// epilogue), which leaves lane l with 8 consecutive columns starting at
// kb(l) = 16 ((l>>4)&1) + 8 (l>>5) -- the weight fragment is simply fetched with the same
// permutation, so both MFMA operands agree on which k every slot carries.  Weights stream
// through LDS: the NWAVE (4) waves of a workgroup (64 rows) fill chunks of <= 32 fragment blocks by
// LDS-DMA, already in MFMA fragment order (1 KB per 16 x 32 block = the 64 lanes' 16-byte
// pieces), into a ring of four 32-KB buffers: every wave issues exactly DPW = 32 / NWAVE DMA instructions
// per chunk (padding goes to a dump block), so "chunk i has landed" is the counted wait
// vmcnt(2 DPW) with chunks i+1 and i+2 still in flight, chunk i+3 is issued behind the barrier
// that frees its buffer.  Each weight byte leaves L2 once per workgroup and every fragment
// read is a conflict-free ds_read_b128.  (Measured: weights from L2 per wave 81 us per chain,
// double-buffered LDS chunks 40 us, this ring: see DESIGN.md.)
//
// Three chains per block replace fifteen launches (5 LayerNorms, 9 GEMMs, 1 memset):
//   A  x += 1/2 FF1(x);  q|k|v = W_qkv LN(x)                          (before the attention)
//   B  x += W_out attn + b;  glu_in = W_pw1 LN(x) + b                 (before the depthwise conv)
//   C  x += W_pw2 u + b;  x += 1/2 FF2(x);  x = LN_post(x)            (end of the block)
// E = 144 (the reference's emb_size), FF = 576, conv inner = 288; other sizes take the
// unfused path of afx_engine.hip.
#include "afx_common.h"
#include "afx_kernels.h"

// Timing experiments only (wrong results): build with -DCHAIN_DBG=<mask> into a separate library
// (1: no weight DMA after the prologue, 2: no MFMA, 4: no barriers, 8: every fragment read hits one LDS address).
#ifndef CHAIN_DBG
#define CHAIN_DBG 0
#endif

namespace afx {

namespace {

constexpr int ET = 9;   // 16-column tiles of the residual stream (E = 144)
constexpr int EK = 5;   // 32-wide k-steps covering E (144 -> 160, the pad columns are zero)
constexpr int HT6 = 6;  // the FF hidden layer is walked in 6 parts of 6 tiles (96 columns = 3 k-steps)
constexpr int RING = 4, BUFSZ = 32 * 1024, DUMP = RING * BUFSZ;  // + one 1-KB dump block
constexpr int PARAMS = DUMP + 1024;  // byte offset of the parameter block behind the ring and the dump block
// Waves per workgroup (16 token rows each).  The chain is a latency path per wave, not a throughput one: at
// B = 64 there are only 796 row tiles, so 8 waves per workgroup fill 100 of the 256 CUs with two waves per SIMD
// contending for the LDS fragment reads, 4 waves fill 199 CUs with one wave per SIMD: 30.9 -> 27.3 us per
// launch, bit-identical results (tools/chain_ab.py; each weight byte is then fetched once per 64 rows
// instead of 128 -- from L2, 0.66 MB per chain, not a cost).
#ifndef CHAIN_NWAVE
#define CHAIN_NWAVE 4
#endif
constexpr int NWAVE = CHAIN_NWAVE;
constexpr int DPW = 32 / NWAVE;     // LDS-DMA instructions per wave per weight chunk
static_assert(NWAVE == 8 || NWAVE == 4, "wave count");
template <int N>
__device__ __forceinline__ void chain_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct ChunkDesc {  // NT x KS fragment blocks of matrix W: rows n0 .. n0 + 16 NT, k-steps k0 .. k0 + KS
  const void* W;
  int ldw, n0, k0, nt, ks;
};

// the static weight-chunk sequence of each chain (must match the compute order below)
// Split precision (S3): the weights are the engine's pair-form rows (hi / lo halves interleaved per 32 k: 2 ldw halfs per row,
// afx_kernels.h::s3_pair_index), a fragment block is 1 KB of hi + 1 KB of lo halves, and a chunk is ONE group of three output
// tiles (<= 15 blocks x 2 = 30 of the 32 DMA slots): every chunk of the fp16 sequence becomes nt / 3 chunks.
template <int STAGE, bool S3>
__device__ __forceinline__ ChunkDesc chunk_of(const ConfChainArgs& p, int idx) {
  if constexpr (S3) {
    auto ff = [&](int c) {  // 30 chunks: per part W1 tiles 0-2, 3-5, then W2 tiles 0-2, 3-5, 6-8 of that part's 3 k-steps
      const int part = c / 5, r = c - 5 * part;
      return r < 2 ? ChunkDesc{p.ff_w1, p.Ep, part * HT6 * 16 + r * 48, 0, 3, EK} : ChunkDesc{p.ff_w2, p.FFp, (r - 2) * 48, 3 * part, 3, 3};
    };
    auto rows3 = [&](const void* W, int ld, int c) { return ChunkDesc{W, ld, c * 48, 0, 3, EK}; };
    if constexpr (STAGE == 0) return idx < 30 ? ff(idx) : rows3(p.w_a, p.Ep, idx - 30);
    else if constexpr (STAGE == 1) return idx < 3 ? rows3(p.w_a, p.Ep, idx) : rows3(p.w_b, p.Ep, idx - 3);
    else return idx < 9 ? ChunkDesc{p.w_a, p.ld_w_a, (idx % 3) * 48, 3 * (idx / 3), 3, 3} : ff(idx - 9);
  } else {
    auto ff = [&](int c) {  // 12 chunks: W1 part, W2 part, ...
      const int part = c >> 1;
      return (c & 1) ? ChunkDesc{p.ff_w2, p.FFp, 0, 3 * part, ET, 3} : ChunkDesc{p.ff_w1, p.Ep, part * HT6 * 16, 0, HT6, EK};
    };
    auto rows6 = [&](const void* W, int ld, int c, int tiles) {  // N = 16 tiles, six tiles per chunk (the last may be shorter)
      const int left = tiles - 6 * c;
      return ChunkDesc{W, ld, c * 96, 0, left < 6 ? left : 6, EK};
    };
    if constexpr (STAGE == 0) return idx < 12 ? ff(idx) : rows6(p.w_a, p.Ep, idx - 12, 3 * ET);
    else if constexpr (STAGE == 1) return idx < 2 ? rows6(p.w_a, p.Ep, idx, ET) : rows6(p.w_b, p.Ep, idx - 2, 4 * ET);
    else return idx < 3 ? ChunkDesc{p.w_a, p.ld_w_a, 0, 3 * idx, ET, 3} : ff(idx - 3);
  }
}
template <int STAGE, bool S3> constexpr int kChunks = S3 ? (STAGE == 0 ? 30 + 9 : STAGE == 1 ? 3 + 12 : 9 + 30)
                                                         : (STAGE == 0 ? 12 + 5 : STAGE == 1 ? 2 + 6 : 3 + 12);

template <class HT, int STAGE, bool S3>
struct Chain {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  // the A operand of one 32-wide k-step: this lane's 8 consecutive k; in split precision their hi and lo halves
  // (x ~ h + l, |l| <= ulp(h) / 2: the product is h.wh + l.wh + h.wl, three matrix-core passes; `l` is untouched otherwise)
  struct AF {
    V8 h, l;
  };
  static constexpr int NCH = kChunks<STAGE, S3>;
  static constexpr int PRM_PIECES = S3 ? 16 : 8;  // 1-KB pieces of the parameter block(s)

  const ConfChainArgs& p;
  int lane, wave, r16, kq, kb;
  char* smem;
  unsigned lds_base;
  int consumed = 0;
  __device__ Chain(const ConfChainArgs& args, char* lds) : p(args) {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    r16 = lane & 15;
    kq = lane >> 4;
    kb = (kq & 1) * 16 + (kq >> 1) * 8;
    smem = lds;
    lds_base = (unsigned)(size_t)lds;
    prm = (const float*)(lds + PARAMS);
    scl = prm + kChainParamFloats;
  }
  const float *prm, *scl;

  // ---- weight stream -------------------------------------------------------------------
  __device__ __forceinline__ void issue(int idx) {  // exactly 32 / NWAVE LDS-DMA instructions per wave
    if constexpr ((CHAIN_DBG & 1) != 0) {
      if (idx >= RING - 1) return;
    }
    const ChunkDesc d = chunk_of<STAGE, S3>(p, idx);
    const int nb = d.nt * d.ks;
    constexpr int KW = S3 ? 64 : 32;  // halfs per k-step in a weight row (pair form: 32 hi + 32 lo)
    const long ldw = S3 ? 2L * d.ldw : d.ldw;
    const unsigned dst0 = lds_base + (idx & (RING - 1)) * BUFSZ;
    const T* src0 = (const T*)d.W + (long)(d.n0 + r16) * ldw + d.k0 * KW + kb;
#pragma unroll
    for (int i = 0; i < 32 / NWAVE; ++i) {
      const int b = i * NWAVE + wave;
      const bool lo = S3 && b >= nb;           // slots nb .. 2 nb - 1: the lo halves of blocks 0 .. nb - 1
      const int blk = lo ? b - nb : b;
      const bool valid = S3 ? b < 2 * nb : b < nb;
      const int j = d.ks == 3 ? (blk * 11) >> 5 : (blk * 13) >> 6;  // blk / ks for blk < 32, ks in {3, 5}
      const int k = blk - j * d.ks;
      const T* src = valid ? src0 + (long)j * 16 * ldw + k * KW + (lo ? 32 : 0) : src0;
      unsigned keep;
      const unsigned dst = __builtin_amdgcn_readfirstlane(valid ? dst0 + b * 1024 : lds_base + DUMP);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  }
  // The per-column vectors (biases, LayerNorm gamma/beta: one packed 8-KB block per chain; split precision: a second
  // 8-KB block with the weight rows' scales) ride the same DMA stream, first in the queue.  Nothing in the chain
  // is a compiler-visible global load: the DMAs are issued from inline asm, so a compiler wait
  // for one of its own loads would be vmcnt(0) and drain the weight ring every time (measured:
  // 36 us per chain with the bias vectors read from global inside the chain).
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int pb = 0; pb < PRM_PIECES / NWAVE; ++pb) {  // 1-KB pieces
      unsigned keep;
      const int piece = pb * NWAVE + wave;
      const float* src = p.params + piece * 256 + lane * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + PARAMS + piece * 1024);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    static_assert(NCH >= RING - 1 && 8 * 256 == kChainParamFloats && 8 * 256 == kChainScaleFloats, "prologue shape");
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue(i);
    chain_wait_vmcnt<3 * DPW>();  // the parameter block (oldest) has landed; 3 chunks stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // Before consuming chunk i: its DMA has landed on every wave (counted wait, then the barrier),
  // and every wave is done with chunk i-1, whose buffer chunk i+3 now takes.  Returns the
  // lane's fragment base inside chunk i.
  __device__ __forceinline__ const char* next_chunk() {
    const int i = consumed++;
    const int after = NCH - 1 - i;  // chunks issued behind this one
    if constexpr ((CHAIN_DBG & 1) == 0) {
      if (after >= 2) chain_wait_vmcnt<2 * DPW>();
      else if (after == 1) chain_wait_vmcnt<DPW>();
      else chain_wait_vmcnt<0>();
    }
    if constexpr ((CHAIN_DBG & 4) == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (i + RING - 1 < NCH) issue(i + RING - 1);
    return smem + (i & (RING - 1)) * BUFSZ + lane * 16;
  }

  // Fragment reads are batched per group of 3 output tiles (3 KS ds_read_b128 in flight) and the
  // MFMAs walk k outside, tiles inside (3 independent accumulators back to back); the second
  // wave of the SIMD fills the read latency.  The sched_barriers pin that shape: left alone,
  // hipcc emits read -> wait -> MFMA on one accumulator at a time.
  template <int NT, int KS, bool ACCUM>
  __device__ __forceinline__ void gemm_impl(const AF* a, const char* wb, f32x4* acc) const {
    static_assert(NT % 3 == 0, "tiles are processed in groups of 3");
    constexpr int NG = NT / 3;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      V8 w[3][KS];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < KS; ++k) w[t][k] = *(const V8*)(wb + ((CHAIN_DBG & 8) ? 0 : ((g * 3 + t) * KS + k) * 1024));
      __builtin_amdgcn_sched_barrier(0);
      f32x4 c[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) c[t] = ACCUM ? acc[g * 3 + t] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if constexpr ((CHAIN_DBG & 2) != 0) asm volatile("" :: "v"(w[t][k]));
          else c[t] = HT::mfma(w[t][k], a[k].h, c[t]);
        }
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[g * 3 + t] = c[t];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // Split precision: ONE chunk = three output tiles, hi blocks (t KS + k) then lo blocks (3 KS + t KS + k).  k outside,
  // the next k-step's six fragments (hi / lo of three tiles) are read under the nine MFMAs of this one -- the chain runs
  // one wave per SIMD, nobody else covers the read latency.  Passes per k-step: wh.ah, wh.al, wl.ah, tiles inside.
  template <int KS, bool ACCUM>
  __device__ __forceinline__ void gemm3_split(const AF* a, const char* wb, f32x4* acc) const {
    V8 wh[2][3], wl[2][3];
    auto load = [&](int buf, int k) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        wh[buf][t] = *(const V8*)(wb + (t * KS + k) * 1024);
        wl[buf][t] = *(const V8*)(wb + ((3 + t) * KS + k) * 1024);
      }
    };
    f32x4 c[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) c[t] = ACCUM ? acc[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    load(0, 0);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k + 1 < KS) load((k + 1) & 1, k + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 3; ++t) c[t] = HT::mfma(wh[k & 1][t], a[k].h, c[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) c[t] = HT::mfma(wh[k & 1][t], a[k].l, c[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) c[t] = HT::mfma(wl[k & 1][t], a[k].h, c[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = c[t];
  }
  // acc[0 .. NT) (=|+=) A . W^T for the next NT output tiles of the weight stream (the chunk(s) are taken here)
  template <int NT, int KS, bool ACCUM = false>
  __device__ __forceinline__ void mm(const AF* a, f32x4* acc) {
    if constexpr (S3) {
#pragma unroll
      for (int g = 0; g < NT / 3; ++g) gemm3_split<KS, ACCUM>(a, next_chunk(), acc + 3 * g);
    } else {
      gemm_impl<NT, KS, ACCUM>(a, next_chunk(), acc);
    }
  }
  // N = 16 TILES output columns of a K = 160 product, six tiles per weight chunk
  template <int TILES>
  __device__ __forceinline__ void gemm_rows6(const AF* a, f32x4* acc) {
#pragma unroll
    for (int c = 0; c < TILES / 6; ++c) mm<6, EK>(a, acc + 6 * c);
    if constexpr (TILES % 6 != 0) mm<TILES % 6, EK>(a, acc + 6 * (TILES / 6));
  }
  // split precision: the accumulators carry the weight rows' power-of-two scales; `off` = the scale of acc[0]'s column 0
  template <int NT>
  __device__ __forceinline__ void descale(f32x4* acc, int off) const {
    if constexpr (S3) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] *= *(const f32x4*)(scl + off + j * 16 + kq * 4);
    }
  }

  // ---- row math ------------------------------------------------------------------------
  // two adjacent accumulator tiles -> this lane's 8 consecutive k of the 32-wide step
  __device__ __forceinline__ AF pack_pair(f32x4 a, f32x4 b) const {
    AF f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
      const float u = __uint_as_float(sw[0]), v = __uint_as_float(sw[1]);
      f.h[r] = (T)u;
      f.h[4 + r] = (T)v;
      if constexpr (S3) {
        f.l[r] = (T)(u - (float)f.h[r]);
        f.l[4 + r] = (T)(v - (float)f.h[4 + r]);
      }
    }
    return f;
  }
  // 8 consecutive k of an operand row as it lies in memory (split precision: fp32 values, split here)
  __device__ __forceinline__ AF load_frag(const void* base, long row, long ld, int k0) const {
    AF f;
    if constexpr (S3) {
      const float* src = (const float*)base + row * ld + k0;
      const f32x4 u = *(const f32x4*)src, v = *(const f32x4*)(src + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f.h[r] = (T)u[r];
        f.h[4 + r] = (T)v[r];
        f.l[r] = (T)(u[r] - (float)f.h[r]);
        f.l[4 + r] = (T)(v[r] - (float)f.h[4 + r]);
      }
    } else {
      f.h = *(const V8*)((const T*)base + row * ld + k0);
    }
    return f;
  }
  __device__ __forceinline__ void layernorm(const f32x4 (&x)[ET], const float* __restrict__ g, const float* __restrict__ b,
                                            f32x4 (&y)[ET]) const {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < ET; ++j) s += (x[j][0] + x[j][1]) + (x[j][2] + x[j][3]);
    const float mean = rows_sum(s) * (1.0f / (16 * ET));
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < ET; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = x[j][r] - mean;
        q = fmaf(d, d, q);
      }
    const float rstd = 1.0f / sqrtf(rows_sum(q) * (1.0f / (16 * ET)) + 1e-5f);
#pragma unroll
    for (int j = 0; j < ET; ++j) {
      const f32x4 gg = *(const f32x4*)(g + j * 16 + kq * 4), bb = *(const f32x4*)(b + j * 16 + kq * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[j][r] = fmaf((x[j][r] - mean) * rstd, gg[r], bb[r]);
    }
  }
  __device__ __forceinline__ void frags_of_rows(const f32x4 (&y)[ET], AF (&a)[EK]) const {
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = pack_pair(y[2 * t], y[2 * t + 1]);
    a[4] = pack_pair(y[8], f32x4{0.f, 0.f, 0.f, 0.f});
  }
  // split precision keeps fp32 accuracy in the activation too: v_exp + v_rcp + one Newton step (<= 1 ulp of the sigmoid)
  static __device__ __forceinline__ float swish_of(float x) {
    if constexpr (S3) {
      const float d = 1.0f + __builtin_amdgcn_exp2f(fminf(x * -1.4426950408889634f, 126.0f));
      float r = __builtin_amdgcn_rcpf(d);
      r = fmaf(fmaf(-d, r, 1.0f), r, r);
      return x * r;
    } else {
      return swish_fast(x);
    }
  }

  // x += 1/2 (W2 swish(W1 LN(x) + b1) + b2): the macaron feed-forward half step (12 weight chunks; split precision: 30)
  __device__ __forceinline__ void feed_forward(f32x4 (&x)[ET]) {
    f32x4 y[ET];
    layernorm(x, prm + CP_FF_G, prm + CP_FF_B, y);
    AF a[EK];
    frags_of_rows(y, a);
    f32x4 o[ET];
#pragma unroll
    for (int j = 0; j < ET; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int part = 0; part < 6; ++part) {
      f32x4 h[HT6];
      mm<HT6, EK>(a, h);
      descale<HT6>(h, CS_FF1 + part * HT6 * 16);
      AF hf[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int c0 = part * HT6 * 16 + t * 32 + kq * 4;
        const f32x4 ba = *(const f32x4*)(prm + CP_FF_B1 + c0), bb = *(const f32x4*)(prm + CP_FF_B1 + c0 + 16);
        f32x4 u = h[2 * t] + ba, v = h[2 * t + 1] + bb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u[r] = swish_of(u[r]);
          v[r] = swish_of(v[r]);
        }
        hf[t] = pack_pair(u, v);
      }
      mm<ET, 3, true>(hf, o);
    }
    descale<ET>(o, CS_FF2);
#pragma unroll
    for (int j = 0; j < ET; ++j) {
      const f32x4 bb = *(const f32x4*)(prm + CP_FF_B2 + j * 16 + kq * 4);
      x[j] += 0.5f * (o[j] + bb);
    }
  }
};

}  // namespace

template <class HT, int STAGE, bool S3>
__global__ __launch_bounds__(64 * NWAVE) void conf_chain_kernel(ConfChainArgs p) {
  typedef Chain<HT, STAGE, S3> C;
  typedef typename C::AF AF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C c(p, smem);
  // every wave takes part in the weight fills and barriers; rows past M are clamped and not stored
  const long row = ((long)blockIdx.x * NWAVE + c.wave) * 16 + c.r16;
  const bool ok = row < p.M;
  const long rc = ok ? row : p.M - 1;

  f32x4 x[ET];
#pragma unroll
  for (int j = 0; j < ET; ++j) x[j] = *(const f32x4*)(p.x_in + rc * p.E + j * 16 + c.kq * 4);
  AF ain[ET];  // stage 1: attention output (5 k-steps), stage 2: depthwise-conv output (9 k-steps)
  if constexpr (STAGE != 0) {
#pragma unroll
    for (int t = 0; t < (STAGE == 1 ? EK : ET); ++t) ain[t] = c.load_frag(p.in_h, rc, p.ld_in_h, t * 32 + c.kb);
  }
  // the row loads are consumed (compiler-placed waits) before the DMA stream starts
#pragma unroll
  for (int j = 0; j < ET; ++j) asm volatile("" : "+v"(x[j]));
  if constexpr (STAGE != 0) {
#pragma unroll
    for (int t = 0; t < (STAGE == 1 ? EK : ET); ++t) {
      asm volatile("" : "+v"(ain[t].h));
      if constexpr (S3) asm volatile("" : "+v"(ain[t].l));
    }
  }
  c.prologue();

  if constexpr (STAGE == 0) {
    c.feed_forward(x);
    if (ok) {
#pragma unroll
      for (int j = 0; j < ET; ++j) *(f32x4*)(p.x_out + row * p.E + j * 16 + c.kq * 4) = x[j];
    }
    f32x4 y[ET];
    c.layernorm(x, c.prm + CP_LN2_G, c.prm + CP_LN2_B, y);
    AF a[EK];
    c.frags_of_rows(y, a);
    // q | k | v: 3 x 144 columns (27 tiles), no bias; six tiles per weight chunk, the last one three
#pragma unroll
    for (int part = 0; part < 5; ++part) {
      f32x4 q[6];
      if (part < 4) c.template mm<6, EK>(a, q);
      else c.template mm<3, EK>(a, q);
      if (part < 4) c.template descale<6>(q, CS_A + part * 96);
      else c.template descale<3>(q, CS_A + part * 96);
      if (ok) {
#pragma unroll
        for (int j = 0; j < (part < 4 ? 6 : 3); ++j) *(f32x4*)(p.out2 + row * p.ld_out2 + (part * 6 + j) * 16 + c.kq * 4) = q[j];
      }
    }
  } else if constexpr (STAGE == 1) {
    // attention output (operand type, row stride ld_in_h, pad columns zero) -> out-projection
    AF a[EK];
#pragma unroll
    for (int t = 0; t < EK; ++t) a[t] = ain[t];
    f32x4 o[ET];
    c.template gemm_rows6<ET>(a, o);
    c.template descale<ET>(o, CS_A);
#pragma unroll
    for (int j = 0; j < ET; ++j) x[j] += o[j] + *(const f32x4*)(c.prm + CP_BA + j * 16 + c.kq * 4);
    if (ok) {
#pragma unroll
      for (int j = 0; j < ET; ++j) *(f32x4*)(p.x_out + row * p.E + j * 16 + c.kq * 4) = x[j];
    }
    f32x4 y[ET];
    c.layernorm(x, c.prm + CP_LN2_G, c.prm + CP_LN2_B, y);
    c.frags_of_rows(y, a);
    // pointwise conv 1 (144 -> 576, bias): the GLU input, fp32; six tiles per weight chunk
#pragma unroll
    for (int part = 0; part < 6; ++part) {
      f32x4 gl[6];
      c.template mm<6, EK>(a, gl);
      c.template descale<6>(gl, CS_B + part * 96);
      if (ok) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int col = part * 96 + j * 16 + c.kq * 4;
          *(f32x4*)(p.out2 + row * p.ld_out2 + col) = gl[j] + *(const f32x4*)(c.prm + CP_BB + col);
        }
      }
    }
  } else {
    // depthwise-conv output (operand type, 288 columns) -> pointwise conv 2 -> residual
    const AF* a = ain;
    f32x4 o[ET];
#pragma unroll
    for (int j = 0; j < ET; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) c.template mm<ET, 3, true>(a + 3 * kc, o);
    c.template descale<ET>(o, CS_A);
#pragma unroll
    for (int j = 0; j < ET; ++j) x[j] += o[j] + *(const f32x4*)(c.prm + CP_BA + j * 16 + c.kq * 4);
    c.feed_forward(x);
    f32x4 y[ET];
    c.layernorm(x, c.prm + CP_LN2_G, c.prm + CP_LN2_B, y);
    if (ok) {
#pragma unroll
      for (int j = 0; j < ET; ++j) *(f32x4*)(p.x_out + row * p.E + j * 16 + c.kq * 4) = y[j];
    }
  }
}

template <class HT, int STAGE, bool S3>
static hipError_t launch_conf_chain_t(const ConfChainArgs& p, hipStream_t s) {
  constexpr int lds = PARAMS + (kChainParamFloats + (S3 ? kChainScaleFloats : 0)) * 4;
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)conf_chain_kernel<HT, STAGE, S3>, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((conf_chain_kernel<HT, STAGE, S3>), dim3((unsigned)((p.M + 16 * NWAVE - 1) / (16 * NWAVE))), dim3(64 * NWAVE), lds, s, p);
  return hipGetLastError();
}

// dtype DT_FP16X3: fp32 operand rows in (in_h), pair-form weights (+ the scale block behind the parameter block), fp32 out
const char* launch_conf_chain(const ConfChainArgs& p, int stage, int dtype, hipStream_t s) {
  if (p.E != 16 * ET || p.Ep < 32 * EK || p.FFp != 4 * 16 * ET) return "conf_chain: the fused Conformer chains are built for emb 144 / ff 576";
  if (dtype == DT_FP32) return "conf_chain: half-precision operands only";
  if (p.M <= 0 || stage < 0 || stage > 2) return "conf_chain: bad arguments";
  hipError_t e;
  if (dtype == DT_FP16X3)
    e = stage == 0 ? launch_conf_chain_t<FP16, 0, true>(p, s) : stage == 1 ? launch_conf_chain_t<FP16, 1, true>(p, s) : launch_conf_chain_t<FP16, 2, true>(p, s);
  else if (dtype == DT_BF16)
    e = stage == 0 ? launch_conf_chain_t<BF16, 0, false>(p, s) : stage == 1 ? launch_conf_chain_t<BF16, 1, false>(p, s) : launch_conf_chain_t<BF16, 2, false>(p, s);
  else
    e = stage == 0 ? launch_conf_chain_t<FP16, 0, false>(p, s) : stage == 1 ? launch_conf_chain_t<FP16, 1, false>(p, s) : launch_conf_chain_t<FP16, 2, false>(p, s);
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace afx
