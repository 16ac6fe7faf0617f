// Matrix-core GEMM for gfx950:  C[m][n] = resid + alpha * act( sum_k A[m][k] * W[n][k] + bias[n] )
//
// One kernel serves every dense product on the path:
//   * conv feature-extractor layers 1-6: with channel-last activations (B,T,512) a
//     k-tap stride-s Conv1d row is the CONTIGUOUS slice in[t*s : t*s+k, :], so the
//     layer is a GEMM with K = k*512 whose A rows overlap (row stride s*512);
//   * post_extract_proj, QKV / out-proj / FC1 / FC2, LL, the Conformer linears;
//   * the grouped positional conv (k=128, 16 groups): K is split in 64-wide chunks,
//     chunk j of output frame t starts at padded frame t+j (kchunk addressing),
//     grid.z walks the groups.
//
// Structure (per 256-thread workgroup = 4 waves in 2x2, BM x BN x 64 tile):
//   global -> LDS by 16-byte LDS-DMA (global_load_lds_dwordx4), two LDS stages, the
//   DMA of K-tile kt+1 in flight under the MFMAs of tile kt, one barrier per K-tile.
//   LDS tiles are [row][64 halfs] = 128-B rows; the 16-B chunk index is XOR-swizzled
//   with (row>>1)&7 so each ds_read_b128 lane group covers 16 distinct slots of the
//   256-B bank row.  LDS-DMA writes lane-linearly, so the swizzle is applied on the
//   per-lane SOURCE address and again on the read (cdna guide rule 21).
//   v_mfma_f32_16x16x32_{bf16,f16}, operands swapped (W fragment as A-operand) so a
//   lane ends up with 4 consecutive output columns of one row -> 16-B stores.
#include "afx_common.h"
#include "afx_kernels.h"

#include <type_traits>

// Timing-only switches of the epilogue (GemmArgs::dbg_nodma: 8 no activation, 16 narrow stores, 32 no stores,
// 64 no epilogue; 8-phase K-loop: 128 / 256 / 512 / 1024 one half-tile's operand DMA off, 2048 no MFMAs, 4096 no LDS
// fragment reads; 8192 the clock-stamping instance (stamps overwrite the head of out_h) -- WRONG results) exist only in the attribution build (make attr -> lib/libafx_attr.so,
// -DAFX_ATTR, loaded through AFX_LIB by tools/bench_convln_attr.py / bench_gemm_k.py); the product library
// compiles them out, so no environment variable or debug key can switch results off.
#ifdef AFX_ATTR
#define AFX_DBG(p, bit) ((p).dbg_nodma & (bit))
#else
#define AFX_DBG(p, bit) 0
#endif

namespace afx {

// ---------------------------------------------------------------------------------------
// Epilogue shared by the tile kernels: acc[i][j] (16x16 tiles of the wave's block, operands
// swapped so a lane holds 4 consecutive columns of one row) -> bias / activation / residual
// / LayerNorm -> wide stores.
// ---------------------------------------------------------------------------------------
// split precision: 8 consecutive results (columns n .. n+7, n % 8 == 0: inside one 32-element group) as the next product's
// A operand -- the pair form of scale x value in place of the fp32 row (GemmArgs::oh_pairs, oh_scale)
__device__ __forceinline__ void store_pairs8(void* out_h, long hrow, long ldo_h, long col, const f32x4& va, const f32x4& vb, float scale) {
  f16x8 hi, lo;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float sa = va[r] * scale, sb = vb[r] * scale;
    hi[r] = (_Float16)sa;
    hi[4 + r] = (_Float16)sb;
    lo[r] = (_Float16)(sa - (float)hi[r]);
    lo[4 + r] = (_Float16)(sb - (float)hi[4 + r]);
  }
  _Float16* hp = (_Float16*)out_h + hrow * (2 * ldo_h) + s3_pair_index(col);
  *(f16x8*)hp = hi;
  *(f16x8*)(hp + 32) = lo;
}

// LEAN (the 8-phase kernels): only what the launcher sends them -- no activation or erf-GELU, N % 8 == 0.
// The general form inlines the other activations at every one of the 32 steps and carries the 4-column
// fallback: ~300 KB of code around a 12-KB K-loop, refetched through the instruction cache after every tile.
// S3 (split precision, GemmArgs::k1): the accumulator of column n is multiplied by pre_scale[n] before the bias, and the
// "operand type" output out_h is written as fp32 (the engine's operand buffers are fp32 in that mode).
template <class HT, int BM, int BN, int WR, int WC, bool ROWLN, bool LEAN = false, bool S3 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[BM / WR / 16][BN / WC / 16], char* smem,
                                              int m0, int n0, int g) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  typedef typename HT::V4 V4;
  constexpr int WM = BM / WR, WN = BN / WC;
  constexpr int MT = WM / 16, NT = WN / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  // epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
  if constexpr (ROWLN) {
    static_assert(BN % (16 * WC) == 0, "row-complete tile");
    __syncthreads();  // every wave is done reading the last K-tile: LDS becomes scratch
    float* red = (float*)smem;  // [WC][BM] partial row sums
    // bias, gamma, beta of the BN columns: one global round trip for the whole workgroup, then LDS
    // (fetched per use they were a dependent L2 round trip per column pair)
    float* vec = red + WC * BM;  // [3][BN] (+ a fourth row, the column scales, in split precision)
    for (int t = tid; t < BN; t += 64 * WR * WC) {
      vec[t] = p.bias[n0 + t];
      vec[BN + t] = p.ln_gamma[n0 + t];
      vec[2 * BN + t] = p.ln_beta[n0 + t];
      if constexpr (S3) vec[3 * BN + t] = p.pre_scale[n0 + t] * p.a_inv;
    }
    __syncthreads();
    const int kq = lane >> 4;
    // v = acc + bias, kept in the accumulator registers
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4 b = *(const f32x4*)(vec + wc * WN + j * 16 + 4 * kq);
      if constexpr (S3) {
        const f32x4 rs = *(const f32x4*)(vec + 3 * BN + wc * WN + j * 16 + 4 * kq);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = acc[i][j] * rs + b;
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] += b;
      }
    }
    float mean[MT], rstd[MT];
    const float invn = 1.0f / (float)BN;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
      s = rows_sum(s);
      if (kq == 0) red[wc * BM + wr * WM + i * 16 + (lane & 15)] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wr * WM + i * 16 + (lane & 15);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < WC; ++c) s += red[c * BM + row];
      mean[i] = s * invn;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[i][j][r] -= mean[i];
          s = fmaf(acc[i][j][r], acc[i][j][r], s);
        }
      s = rows_sum(s);
      if (kq == 0) red[wc * BM + wr * WM + i * 16 + (lane & 15)] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wr * WM + i * 16 + (lane & 15);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < WC; ++c) s += red[c * BM + row];
      rstd[i] = 1.0f / sqrtf(s * invn + p.ln_eps);
    }
    // normalise, activate, then widen to 8 columns per lane (v_permlane16_swap, see below)
    const int cbw = (kq & 1) * 16 + (kq >> 1) * 8;
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
      const int ca = wc * WN + jp * 32 + 4 * kq, cbb = ca + 16;  // columns inside the tile
      const f32x4 ga0 = *(const f32x4*)(vec + BN + ca), be0 = *(const f32x4*)(vec + 2 * BN + ca);
      const f32x4 ga1 = *(const f32x4*)(vec + BN + cbb), be1 = *(const f32x4*)(vec + 2 * BN + cbb);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + wr * WM + i * 16 + (lane & 15);
        f32x4 va, vb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          va[r] = fmaf(acc[i][2 * jp][r] * rstd[i], ga0[r], be0[r]);
          vb[r] = fmaf(acc[i][2 * jp + 1][r] * rstd[i], ga1[r], be1[r]);
        }
        if (p.act == ACT_GELU && !(AFX_DBG(p, 8))) {
          if constexpr (S3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              va[r] = gelu_erf(va[r]);
              vb[r] = gelu_erf(vb[r]);
            }
          } else {
            gelu_poly8(va, vb);
          }
        } else if (!LEAN && p.act != ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            va[r] = apply_act(va[r], p.act);
            vb[r] = apply_act(vb[r], p.act);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va[r]), __float_as_uint(vb[r]), false, false);
          va[r] = __uint_as_float(sw[0]);
          vb[r] = __uint_as_float(sw[1]);
        }
        if (AFX_DBG(p, 32)) {  // timing only: no stores
          asm volatile("" :: "v"(va), "v"(vb));
          continue;
        }
        if (m >= p.M) continue;
        const int n = n0 + wc * WN + jp * 32 + cbw;
        const long orow = (long)(m / p.rpb) * p.o_batch_rows + (m % p.rpb) + p.o_row_off;
        const long hrow = (long)(m / p.rpb) * p.oh_batch_rows + (m % p.rpb) + p.oh_row_off;
        if (p.out_f) {
          float* op = p.out_f + orow * p.ldo_f + n;
          *(f32x4*)op = va;
          *(f32x4*)(op + 4) = vb;
        }
        if (p.out_h) {
          if constexpr (S3) {
            if (p.oh_pairs) {
              store_pairs8(p.out_h, hrow, p.ldo_h, n, va, vb, p.oh_scale);
            } else {
              float* op = (float*)p.out_h + hrow * p.ldo_h + n;
              *(f32x4*)op = va;
              *(f32x4*)(op + 4) = vb;
            }
          } else {
            V8 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              h[r] = (T)va[r];
              h[4 + r] = (T)vb[r];
            }
            *(V8*)((T*)p.out_h + hrow * p.ldo_h + n) = h;
          }
        }
      }
    }
    return;
  }
  const int gcol = g * p.g_n;
  const float alpha = p.alpha;
  const int kq = lane >> 4;
  if (LEAN || ((p.N & 7) == 0 && !(AFX_DBG(p, 16)))) {
    // Wide-store epilogue.  After the MFMAs a lane holds 4 consecutive columns of one row
    // (8 B of fp16); v_permlane16_swap exchanges 16-lane rows between the registers of two
    // adjacent 16-column tiles so that every lane ends up with 8 consecutive columns:
    //   row-of-lane 0: tile j cols 0-7 | 1: tile j+1 cols 0-7 | 2: tile j cols 8-15 | 3: tile j+1 cols 8-15
    // -> one 16-B store per lane (two for fp32), half the store instructions of the narrow
    // form; the store tail of these kernels is issue-bound (cdna guide T21).
    const int cb = (kq & 1) * 16 + (kq >> 1) * 8;  // column base of this lane inside a tile pair
    // Every global load of the epilogue is issued ahead of its use: the bias vectors (they do not
    // depend on the row) before the loop, the residual rows one (row tile, column pair) step ahead.
    // Loaded at the point of use, each of the MT x NT/2 steps was a dependent L2 round trip --
    // ~15 us per 256x256 tile, more than its K-loop at K = 512 (tools/bench_gemm_k.py).
    f32x4 bia[NT / 2][2];
    f32x4 rsc[S3 ? NT / 2 : 1][2];  // split precision: the column scales, loaded like the bias
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
      const int na = n0 + wc * WN + jp * 32 + 4 * kq, nbb = na + 16;
      bia[jp][0] = bia[jp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        if (na < p.N) bia[jp][0] = *(const f32x4*)(p.bias + gcol + na);
        if (nbb < p.N) bia[jp][1] = *(const f32x4*)(p.bias + gcol + nbb);
      }
      if constexpr (S3) {
        rsc[jp][0] = rsc[jp][1] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (na < p.N) rsc[jp][0] = *(const f32x4*)(p.pre_scale + gcol + na) * p.a_inv;
        if (nbb < p.N) rsc[jp][1] = *(const f32x4*)(p.pre_scale + gcol + nbb) * p.a_inv;
      }
    }
    constexpr int STEPS = MT * (NT / 2);
    const bool one_batch = p.rpb >= p.M;  // plain GEMM: no per-batch row remap, no integer divisions
    auto row_of = [&](int i, long& orow, long& hrow, bool& mok) {
      const int m = m0 + wr * WM + i * 16 + (lane & 15);
      mok = m < p.M;
      const int mc = mok ? m : p.M - 1;
      const int bq = one_batch ? 0 : mc / p.rpb, br = one_batch ? mc : mc - bq * p.rpb;
      orow = (long)bq * p.o_batch_rows + br + p.o_row_off;
      hrow = (long)bq * p.oh_batch_rows + br + p.oh_row_off;
    };
    auto load_resid = [&](int step, f32x4 (&r)[2]) {
      const int i = step / (NT / 2), jp = step % (NT / 2);
      long orow, hrow;
      bool mok;
      row_of(i, orow, hrow, mok);
      const int n = n0 + wc * WN + jp * 32 + cb;
      r[0] = r[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (mok && n < p.N) {
        const float* rp = p.resid + orow * p.ldr + gcol + n;
        r[0] = *(const f32x4*)rp;
        r[1] = *(const f32x4*)(rp + 4);
      }
    };
    // One step = (row tile i, column pair jp).  HAS_R is a compile-time copy of "p.resid != nullptr" so that the
    // no-residual form contains no load at all: on gfx9 stores and loads share vmcnt and return out of order with
    // respect to each other, so ANY wait for a load issued before a store is vmcnt(0) = wait for the store's
    // acknowledgement.  The one-step-ahead residual prefetch of the first version (and the unconditional
    // register copy behind it) cost exactly that at each of the 16 steps, residual or not (~1 us each, the bulk
    // of the exposed epilogue time).  The residual rows now come in chunks of RCH steps, loaded together
    // before the chunk's first store: STEPS / RCH = 8 waits per tile instead of 16, and none without a residual.
    auto do_step = [&](int step, auto has_r, const f32x4 (&rr)[2]) {
      constexpr bool HAS_R = decltype(has_r)::value;
      const int i = step / (NT / 2), jp = step % (NT / 2);
      long orow, hrow;
      bool mok;
      row_of(i, orow, hrow, mok);
      const int nb = n0 + wc * WN + jp * 32;  // first column of the tile pair
      f32x4 va, vb;
      if constexpr (S3) {
        va = acc[i][2 * jp] * rsc[jp][0] + bia[jp][0];
        vb = acc[i][2 * jp + 1] * rsc[jp][1] + bia[jp][1];
      } else {
        va = acc[i][2 * jp] + bia[jp][0];
        vb = acc[i][2 * jp + 1] + bia[jp][1];
      }
      if (p.act == ACT_GELU && !(AFX_DBG(p, 8))) {
        if constexpr (S3) {  // fp32 results: the fp32-accurate erf form (the polynomial is sized for fp16 outputs)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            va[r] = gelu_erf(va[r]);
            vb[r] = gelu_erf(vb[r]);
          }
        } else {
          gelu_poly8(va, vb);
        }
      } else if (!LEAN && p.act != ACT_NONE && !(AFX_DBG(p, 8))) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          va[r] = apply_act(va[r], p.act);
          vb[r] = apply_act(vb[r], p.act);
        }
      }
      if (alpha != 1.f) {
        va *= alpha;
        vb *= alpha;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va[r]), __float_as_uint(vb[r]), false, false);
        va[r] = __uint_as_float(sw[0]);
        vb[r] = __uint_as_float(sw[1]);
      }
      const int n = nb + cb;  // this lane now owns columns n .. n+7 (va | vb)
      if (AFX_DBG(p, 32)) {  // timing only: no stores
        asm volatile("" :: "v"(va), "v"(vb));
      } else if (mok && n < p.N) {
        if (HAS_R) {
          va += rr[0];
          vb += rr[1];
        }
        if (p.out_f) {
          float* op = p.out_f + orow * p.ldo_f + gcol + n;
          *(f32x4*)op = va;
          *(f32x4*)(op + 4) = vb;
        }
        if (p.out_h) {
          if constexpr (S3) {
            if (p.oh_pairs) {  // the next product's A operand, in pair form
              store_pairs8(p.out_h, hrow, p.ldo_h, gcol + n, va, vb, p.oh_scale);
            } else {
              float* op = (float*)p.out_h + hrow * p.ldo_h + gcol + n;
              *(f32x4*)op = va;
              *(f32x4*)(op + 4) = vb;
            }
          } else {
            V8 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              h[r] = (T)va[r];
              h[4 + r] = (T)vb[r];
            }
            *(V8*)((T*)p.out_h + hrow * p.ldo_h + gcol + n) = h;
          }
        }
      }
    };
    if (p.resid) {
      constexpr int RCH = STEPS % 2 == 0 ? 2 : 1;  // 2 steps = 16 registers of residual in flight: 4 already spill the 256x256 tile (245 VGPRs)
#pragma unroll
      for (int c0 = 0; c0 < STEPS; c0 += RCH) {
        f32x4 rr[RCH][2];
#pragma unroll
        for (int u = 0; u < RCH; ++u) load_resid(c0 + u, rr[u]);
#pragma unroll
        for (int u = 0; u < RCH; ++u) do_step(c0 + u, std::true_type{}, rr[u]);
      }
    } else {
      const f32x4 none[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int step = 0; step < STEPS; ++step) do_step(step, std::false_type{}, none);
    }
    return;
  }
  // narrow fallback (N % 8 != 0): 4 columns per lane
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wr * WM + i * 16 + (lane & 15);
    if (m >= p.M) continue;
    const long orow = (long)(m / p.rpb) * p.o_batch_rows + (m % p.rpb) + p.o_row_off;
    const long hrow = (long)(m / p.rpb) * p.oh_batch_rows + (m % p.rpb) + p.oh_row_off;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wc * WN + j * 16 + 4 * kq;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if constexpr (S3) v *= *(const f32x4*)(p.pre_scale + gcol + n) * p.a_inv;
      if (p.bias) v += *(const f32x4*)(p.bias + gcol + n);
      if (p.act != ACT_NONE && !(AFX_DBG(p, 8))) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      }
      v *= alpha;
      if (p.resid) v += *(const f32x4*)(p.resid + orow * p.ldr + gcol + n);
      if (p.out_f) *(f32x4*)(p.out_f + orow * p.ldo_f + gcol + n) = v;
      if (p.out_h) {
        if constexpr (S3) {
          *(f32x4*)((float*)p.out_h + hrow * p.ldo_h + gcol + n) = v;
        } else {
          V4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (T)v[r];
          *(V4*)((T*)p.out_h + hrow * p.ldo_h + gcol + n) = h;
        }
      }
    }
  }
}

// Split precision, one 64-half K-tile of pair-form operands (32 k values): acc += w_lo.a_hi + w_hi.a_lo + w_hi.a_hi, the small
// terms first.  EVERY tile kernel issues exactly this sequence per K-tile and output element, so a row's bits do not depend on
// the tile family that computed it (ragged batches, tests/test_gpu_models.py::test_ragged_bit_identity_holds_across_tile_families).
template <class HT, int MT, int NT>
__device__ __forceinline__ void s3_mfma(const typename HT::V8 (&wh)[NT], const typename HT::V8 (&wl)[NT], const typename HT::V8 (&ah)[MT],
                                        const typename HT::V8 (&al)[MT], f32x4 (&acc)[MT][NT]) {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(wl[j], ah[i], acc[i][j]);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(wh[j], al[i], acc[i][j]);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(wh[j], ah[i], acc[i][j]);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  // beyond the table: the raw encoding (gfx9: vmcnt = bits 3:0 and 15:14, expcnt / lgkmcnt fields at their maximum = no wait)
  if constexpr (N > 10) asm volatile("s_waitcnt %0" ::"n"((N & 15) | ((N >> 4) << 14) | 0x0F70) : "memory");
  else if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}

// WR x WC waves per workgroup; each wave owns a (BM/WR) x (BN/WC) block of the tile.
// ROWLN: the tile spans the whole output row (BN == N), and the epilogue applies
// LayerNorm over the row (two-pass fp32 statistics, partial sums exchanged through LDS
// between the WC waves of a row) followed by the activation -- the conv feature
// extractor's "conv -> LayerNorm(512) -> GELU" in one kernel, no fp32 round trip.
template <class HT, int BM, int BN, int WR, int WC, bool ROWLN = false, bool LEAN = false, bool S3 = false>
__global__ __launch_bounds__(64 * WR * WC) void gemm_kernel(GemmArgs p) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  typedef typename HT::V4 V4;
  constexpr int NW = WR * WC;
  constexpr int WM = BM / WR, WN = BN / WC;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int AI = BM / (8 * NW), WI = BN / (8 * NW);  // LDS-DMA instructions per thread per tile
  static_assert(AI >= 1 && WI >= 1, "tile too small for the wave count");
  constexpr bool ASM_DMA = ROWLN;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int g = blockIdx.z;
  // ---- workgroup -> tile mapping (speed only; any mapping is correct) ----------------
  // Workgroups are dealt round-robin over the 8 XCDs (private 4-MB L2 each).  map 1/2
  // give every XCD a contiguous run of the logical tile order, so tiles that share an A
  // row-panel (all N-tiles of one M-tile) or a W panel meet in ONE L2 instead of being
  // re-fetched by up to 8 of them; map 2 additionally walks the tiles in GROUP_M x nN
  // super-tiles so both panels of the working set stay L2-resident.
  int pm, pn;
  {
    const int nN = (p.N + BN - 1) / BN, nM = (p.M + BM - 1) / BM;
    const int nwg = nM * nN;
    int L = blockIdx.x;
    if (p.map_mode >= 1) {
      const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);  // bijective for any nwg
    }
    if (p.map_mode == 2) {
      constexpr int GM = 8;
      const int width = GM * nN, grp = L / width, first = grp * GM;
      const int gsz = nM - first < GM ? nM - first : GM;
      pm = first + (L % width) % gsz;
      pn = (L % width) / gsz;
    } else {
      pm = L / nN;
      pn = L % nN;
    }
  }
  const int m0 = pm * BM, n0 = pn * BN;

  const T* Ag = (const T*)p.A + (long)g * p.g_a;
  const T* Wg = (const T*)p.W + (long)g * p.g_w;

  // per-lane source pointers (k = 0) of the chunks this lane DMAs each K-tile
  const T* a_src[AI];
  const T* w_src[WI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    a_src[i] = Ag + (long)(m / p.rpb) * p.a_batch + (long)(m % p.rpb) * p.a_row + c * 8;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < p.N ? n : p.N - 1;
    w_src[i] = Wg + (long)n * p.ldw + c * 8;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K >> 6;

  // ASM_DMA (one workgroup per CU, so a wave must overlap its own LDS reads with its MFMAs;
  // measured +20 % on the row-complete conv tile, -5 % on the 2-workgroup-per-CU 128x128 tile):
  // LDS-DMA through inline asm: with the builtin in the loop hipcc's waitcnt pass stops
  // counting lgkmcnt (every ds_read group is followed by lgkmcnt(0), so a wave never overlaps
  // its LDS reads with its MFMAs); hidden in asm, the compiler keeps fine-grained counted
  // waits for the fragment reads, and the DMA's own completion is waited by hand (vmcnt(0)
  // at the top of each K-tile).  M0 (the LDS destination base) is saved and restored inside
  // the statement, as the cdna guide prescribes.
  const unsigned lds_base = (unsigned)(size_t)smem;
  auto dma16 = [&](const T* src, unsigned lds_off, bool nt) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + lds_off);
    if (nt)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto stage = [&](int buf, int kt) {
    // (split precision: the operands are in pair form and K counts its halfs -- the same walk as a plain fp16 operand)
    const int k0 = kt << 6;
    const long ka = (long)(k0 / p.kchunk) * p.kchunk_stride + (k0 % p.kchunk);
    const unsigned base = (unsigned)(buf * STAGE);
    // a_nt: the A panel is read by exactly one workgroup (row-complete tile) -- stream it
    // non-temporally so it does not evict the W panel every workgroup re-reads from L2
    if constexpr (ASM_DMA) {
#pragma unroll
      for (int i = 0; i < AI; ++i) dma16(a_src[i] + ka, base + (i * NW + wave) * 1024, p.a_nt != 0);
#pragma unroll
      for (int i = 0; i < WI; ++i) dma16(w_src[i] + k0, base + A_BYTES + (i * NW + wave) * 1024, false);
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + ka),
                                         (__attribute__((address_space(3))) void*)(smem + base + (i * NW + wave) * 1024),
                                         16, 0, 0);
#pragma unroll
      for (int i = 0; i < WI; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + k0),
                                         (__attribute__((address_space(3))) void*)(smem + base + A_BYTES + (i * NW + wave) * 1024),
                                         16, 0, 0);
    }
  };

  // fragment read offsets (bytes) inside a stage; the swizzle term only depends on lane
  const int frow = lane & 15;
  const int fsw = (frow >> 1) & 7;
  const int a_off = (wr * WM + frow) * 128;
  const int w_off = A_BYTES + (wc * WN + frow) * 128;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; every wave is done with tile kt-1
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sb = smem + (kt & 1) * STAGE;
    if constexpr (S3) {
      // pair form: k-step 0 of the LDS rows holds the hi halves of 32 k values, k-step 1 their lo halves
      V8 af[2][MT], wf[2][NT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int slot = ((ks * 4 + (lane >> 4)) ^ fsw) * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[ks][j] = *(const V8*)(sb + w_off + j * 16 * 128 + slot);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[ks][i] = *(const V8*)(sb + a_off + i * 16 * 128 + slot);
      }
      __builtin_amdgcn_s_setprio(1);
      s3_mfma<HT, MT, NT>(wf[0], wf[1], af[0], af[1], acc);
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int slot = ((ks * 4 + (lane >> 4)) ^ fsw) * 16;
        V8 af[MT], wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *(const V8*)(sb + w_off + j * 16 * 128 + slot);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const V8*)(sb + a_off + i * 16 * 128 + slot);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }

  gemm_epilogue<HT, BM, BN, WR, WC, ROWLN, LEAN, S3>(p, acc, smem, m0, n0, g);
}


// =======================================================================================
// 8-phase 256x256 tile kernel (cdna guide, "The 256^2 8-phase template"): 8 waves as 2(M) x
// 4(N), each owning 128 x 64 outputs; one workgroup per CU, 128 KB of LDS = 2 K-tile buffers
// x 4 half-tiles (A0 A1 B0 B1, 128 rows x 64 halfs each).  Half h of A holds rows
// {64h .. 64h+63} of BOTH wave rows, half h of B holds columns {32h .. 32h+31} of all FOUR
// wave columns, so that "quadrant (Ai, Bj) of every wave" needs exactly half-tiles Ai and Bj.
//
// A K-tile is 4 phases (one 64 x 32 output quadrant x K = 64 each = 16 MFMAs per wave):
//     phase 1: read B0 (4 ds_read_b128) then A0 (8)   stage A1(t+1)   MFMA (A0,B0)
//     phase 2: read B1 (4)                            stage B0(t+2)   MFMA (A0,B1)
//     phase 3: read A1 (8)                            stage A0(t+2)   MFMA (A1,B1)
//     phase 4: --                                     stage B1(t+2)   MFMA (A1,B0)   [B0 kept]
// each phase = { ds_reads ; 2 LDS-DMA ; s_barrier ; lgkmcnt(0) ; 16 MFMA ; s_barrier }.
// The operand DMA is retired ONCE per K-tile, in phase 4, with vmcnt(6): the three half-tiles
// of tile t+2 stay in flight across the barriers, all of tile t+1 has landed and is read from
// the next phase on.  The two wave rows run one barrier apart (wr == 1 takes an extra
// s_barrier up front, wr == 0 at the end), so on every SIMD one wave is in its MFMA segment
// while the other issues its LDS reads and DMA.
// Hazards (all counted in barrier intervals, with the one-interval skew between wave rows):
//   RAW  a wave waits for its own DMA (vmcnt) BEFORE the first barrier of phase 4; readers
//        touch that buffer in the next phase, i.e. behind a barrier every waiter has reached.
//   WAR  a half-tile is re-staged two phases after its last read (A0: read ph1, staged ph3;
//        B1: ph2 -> ph4; A1: ph3 -> ph1 of the next tile), or one phase after when the reads
//        were retired before the reading phase's first barrier (B0: ph1 reads are issued
//        first and retired by lgkmcnt(8) there -> staged in ph2).
// =======================================================================================
// MF (256x256 instance only) = 16-row fragments a wave row actually computes, 5..8: the tile then covers 32*MF rows
// (160 / 192 / 224 / 256) of the SAME LDS layout -- half-tile slots stay 64 rows per wave row, the first
// MF0 = ceil(MF/2) / MF1 = MF - MF0 fragments of them are live, the other slots receive clamped duplicate rows nobody
// reads.  The launcher picks the height whose tile count fills ONE round of the CUs (M = 64 x 199 rows, N = 1024:
// 228 tiles of 224 rows instead of 200 of 256; M = 16 x 199, N = 3072: 240 tiles of 160 rows instead of 156), so a
// sub-round product pays 7/8 or 5/8 of the K-loop instead of idling a fifth to a third of the chip.  Per-row results
// do not depend on the height (same K order).
// TS (attribution build only): every wave stamps the shader clock at five points of each phase of the FIRST output tile
// (start of the read part / DMA landed + reads retired / first barrier passed / MFMAs issued / second barrier passed) into
// 16 KB of LDS behind the operand buffers; workgroup 0 dumps them over the head of out_h at the end (tools/kloop_timeline.py).
template <class HT, int BM, int BN, bool ROWLN, int MF = BM / 32, int PH = 1, bool TS = false, bool S3 = false>
__global__ __launch_bounds__(512) void gemm8_kernel(GemmArgs p) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  // Two instances: 256x256 (A flows, both B halves stay in registers) and the row-complete
  // 128x512 tile of the conv stack (roles swapped: B flows, both A halves stay resident).
  // "R" = the resident operand (4 fragment reads per half), "F" = the flowing one (8 per half):
  //     phase 1: read R0 then F0   stage F1(t+1)   MFMA (F0,R0)
  //     phase 2: read R1           stage R0(t+2)   MFMA (F0,R1)
  //     phase 3: read F1           stage F0(t+2)   MFMA (F1,R1)
  //     phase 4: --                stage R1(t+2)   MFMA (F1,R0)
  constexpr bool WIDE = BN > BM;              // A is the resident operand
  constexpr int RA = BM / 2, RB = BN / 2;     // rows of an A / B half-tile
  // v_mfma_f32_16x16x32: 16-row fragments, 2 k-steps of 32 per K-tile (same FLOP per clock as 32x32x16 -- 16 against
  // 32 cycles per instruction; its 16-MFMA segments of 256 cycles are what the two wave rows alternate on)
  constexpr int FR = 16, KSN = 2;
  constexpr int MTH = RA / 2 / FR, NTH = RB / 4 / FR;  // fragments per half per wave
  constexpr int DA = RA / 64, DB = RB / 64;   // LDS-DMA instructions per thread per half-tile
  constexpr int OFF_A0 = 0, OFF_A1 = RA * 128, OFF_B0 = 2 * RA * 128, OFF_B1 = OFF_B0 + RB * 128;
  constexpr int BUF = 2 * (RA + RB) * 128;
  static_assert((WIDE ? MTH : NTH) * KSN == 4 && (WIDE ? NTH : MTH) * KSN == 8, "resident operand: 4 reads per half, flowing: 8");
  static_assert(2 * (WIDE ? DA : DB) + (WIDE ? DB : DA) == 6, "vmcnt(6) leaves R0, F0, R1 of tile t+2 in flight");
  static_assert(MF == BM / 32 || (!WIDE && MF >= 5 && MF < 8) || (WIDE && MF >= 2 && MF < 4), "short tiles: 160..224 of 256 rows, 64 / 96 of 128");
  constexpr int MF0 = (MF + 1) / 2, MF1 = MF - MF0;  // live fragments of A half 0 / 1
  constexpr int BMC = (MF0 + MF1) * 32;                                       // rows the tile covers
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = blockIdx.z;
  const T* Ag = (const T*)p.A + (long)g * p.g_a;
  const T* Wg = (const T*)p.W + (long)g * p.g_w;
  const int nN = (p.N + BN - 1) / BN, nM = (p.M - p.m_lo + BMC - 1) / BMC;  // the launch covers rows [m_lo, M)
  const int nwg = nM * nN;
  const int nk = p.K >> 6;

  // DMA source pointers.  Piece (i*8 + wave) of a half-tile is LDS rows 8(i*8+wave) .. +7,
  // lane l -> row l>>3, 16-B slot l&7 holding logical chunk (l&7) ^ ((row>>1)&7).
  // A half h, LDS row r: wave row r / (RA/2), row r % (RA/2) of that wave's half h;
  // B half h, LDS row r: wave column r / (RB/4), column r % (RB/4) of that wave's half h.
  const T* srcA[2][DA];
  const T* srcB[2][DB];
  int m0 = 0, n0 = 0;
  // PERSISTENT: the grid is one workgroup per CU and a workgroup walks tiles v = blockIdx.x,
  // + gridDim.x, ...  Workgroups are dealt round-robin over the 8 XCDs and gridDim.x is a
  // multiple of 8, so v & 7 is the XCD for every tile of a workgroup and the remap below (a
  // contiguous run of the logical tile order per XCD, GROUP_M super-tiles) holds as before.
  auto setup = [&](int v) {
    int pm, pn;
    int L = v;
    if (p.map_mode >= 1) {
      const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    if (p.map_mode == 2) {
      constexpr int GM = 8;
      const int width = GM * nN, grp = L / width, first = grp * GM;
      const int gsz = nM - first < GM ? nM - first : GM;
      pm = first + (L % width) % gsz;
      pn = (L % width) / gsz;
    } else {
      pm = L / nN;
      pn = L % nN;
    }
    m0 = p.m_lo + pm * BMC;
    n0 = pn * BN;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < DA; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int m = m0 + (r / (RA / 2)) * (BMC / 2) + h * (MF0 * FR) + r % (RA / 2);
        m = m < p.M ? m : p.M - 1;
        srcA[h][i] = Ag + (long)(m / p.rpb) * p.a_batch + (long)(m % p.rpb) * p.a_row + c * 8;
      }
#pragma unroll
      for (int i = 0; i < DB; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int n = n0 + (r / (RB / 4)) * (BN / 4) + h * (RB / 4) + r % (RB / 4);
        n = n < p.N ? n : p.N - 1;
        srcB[h][i] = Wg + (long)n * p.ldw + c * 8;
      }
    }
  };
  const unsigned lds_base = (unsigned)(size_t)smem;
  auto dma16 = [&](const T* src, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  // element offset of K-tile kt inside an A / W row (split precision: pair-form operands, K counts halfs -- the same walk)
  auto koffA = [&](int kt) -> long { return (long)kt * 64; };
  auto koffB = [&](int kt) -> long { return (long)kt * 64; };
  auto stageA = [&](int h, int buf, int kt) {
    const long ko = koffA(kt);
#pragma unroll
    for (int i = 0; i < DA; ++i) dma16(srcA[h][i] + ko, buf * BUF + (h ? OFF_A1 : OFF_A0) + (i * 8 + wave) * 1024);
  };
  auto stageB = [&](int h, int buf, int kt) {
    const long ko = koffB(kt);
#pragma unroll
    for (int i = 0; i < DB; ++i) dma16(srcB[h][i] + ko, buf * BUF + (h ? OFF_B1 : OFF_B0) + (i * 8 + wave) * 1024);
  };

  // fragment read addresses (bytes): row (lane & (FR-1)) of an FR-row tile; the lane's 8 k-values of k-step
  // ks are logical chunk ks * (64 / 8 / KSN) + (lane / FR)
  const int frow = lane & (FR - 1), fsw = (frow >> 1) & 7, kq = lane / FR;
  int slot[KSN];
#pragma unroll
  for (int ks = 0; ks < KSN; ++ks) slot[ks] = ((ks * (8 / KSN) + kq) ^ fsw) * 16;
  const char* aR = smem + (wr * (RA / 2) + frow) * 128;
  const char* bR = smem + (wc * (RB / 4) + frow) * 128;

  f32x4 acc[MF0 + MF1][2 * NTH];
  // the resident operand keeps both halves in registers, the flowing one a single half
  V8 af[WIDE ? 2 : 1][MTH][KSN], wf[WIDE ? 1 : 2][NTH][KSN];

  auto readA = [&](int buf, int h) {
#pragma unroll
    for (int mi = 0; mi < MTH; ++mi)
      if (mi < (h ? MF1 : MF0)) {
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
          af[WIDE ? h : 0][mi][ks] = *(const V8*)(aR + buf * BUF + (h ? OFF_A1 : OFF_A0) + mi * (FR * 128) + slot[ks]);
      }
  };
  auto readB = [&](int buf, int h) {
#pragma unroll
    for (int nj = 0; nj < NTH; ++nj)
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks)
        wf[WIDE ? 0 : h][nj][ks] = *(const V8*)(bR + buf * BUF + (h ? OFF_B1 : OFF_B0) + nj * (FR * 128) + slot[ks]);
  };
  auto quadrant = [&](int ah, int bh) {  // 16 MFMAs: (A half ah) x (B half bh) x K = 64
    if (AFX_DBG(p, 2048)) return;  // timing only: K-loop without its MFMAs
    __builtin_amdgcn_s_setprio(1);
    // (split precision, pair-form operands: k-step 0 = the hi halves of the K-tile's 32 k values, k-step 1 = their lo halves;
    // three passes w_lo.a_hi, w_hi.a_lo, w_hi.a_hi in the order of s3_mfma -- 24 MFMAs where the fp16 walk issues 16)
#pragma unroll
    for (int ks = 0; ks < (S3 ? 3 : KSN); ++ks) {
      const int kw = S3 ? (ks == 0 ? 1 : 0) : ks, ka = S3 ? (ks == 1 ? 1 : 0) : ks;
#pragma unroll
      for (int mi = 0; mi < MTH; ++mi)
        if (mi < (ah ? MF1 : MF0)) {
#pragma unroll
          for (int nj = 0; nj < NTH; ++nj) {
            acc[ah * MF0 + mi][bh * NTH + nj] =
                HT::mfma(wf[WIDE ? 0 : bh][nj][kw], af[WIDE ? ah : 0][mi][ka], acc[ah * MF0 + mi][bh * NTH + nj]);
          }
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- PH == 3 ("ring3", the DEFAULT of the 256-wide instance; 0-4.5 % over the two-buffer form,
  // profiles/r02_gemm_ring3_ab.txt): all 160 KB of LDS -- the resident operand (B) double-buffered as
  // before, the flowing one (A) in a ring of THREE K-tile buffers, so that A of tile t+2 can go out a whole phase
  // earlier (its slot was last read by tile t-1) and the DMA splits 4 + 4 over the two read intervals instead of 2 + 6;
  // one DMA wait per K-tile (end of phase B), none in phase A.
  //     LDS: [B buf 0 | B buf 1 | A buf 0 | A buf 1 | A buf 2], 32 KB each
  //     phase A: read B0 B1 A0(t)   stage A0 A1 (t+2)   MFMA (A0,B0) (A0,B1)
  //     phase B: read A1(t)         stage B0 B1 (t+2)   wait: tile t+1 landed   MFMA (A1,B1) (A1,B0)
  constexpr int RSZ3 = 2 * RB * 128, FSZ3 = 2 * RA * 128;
  auto stageA3 = [&](int h, int fboff, int kt) {
    const long ko = koffA(kt);
#pragma unroll
    for (int i = 0; i < DA; ++i) dma16(srcA[h][i] + ko, 2 * RSZ3 + fboff + h * (RA * 128) + (i * 8 + wave) * 1024);
  };
  auto stageB3 = [&](int h, int rb, int kt) {
    const long ko = koffB(kt);
#pragma unroll
    for (int i = 0; i < DB; ++i) dma16(srcB[h][i] + ko, rb * RSZ3 + h * (RB * 128) + (i * 8 + wave) * 1024);
  };
  auto readA3 = [&](int fboff, int h) {
#pragma unroll
    for (int mi = 0; mi < MTH; ++mi)
      if (mi < (h ? MF1 : MF0)) {
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
          af[0][mi][ks] = *(const V8*)(aR + 2 * RSZ3 + fboff + h * (RA * 128) + mi * (FR * 128) + slot[ks]);
      }
  };
  auto readB3 = [&](int rb, int h) {
#pragma unroll
    for (int nj = 0; nj < NTH; ++nj)
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks)
        wf[h][nj][ks] = *(const V8*)(bR + rb * RSZ3 + h * (RB * 128) + nj * (FR * 128) + slot[ks]);
  };
  // R / F views of the two operands
  // (AFX_DBG 4096, timing only: K-loop without its LDS fragment reads)
  auto readR = [&](int buf, int h) { if (AFX_DBG(p, 4096)) return; if constexpr (WIDE) readA(buf, h); else readB(buf, h); };
  auto readF = [&](int buf, int h) { if (AFX_DBG(p, 4096)) return; if constexpr (WIDE) readB(buf, h); else readA(buf, h); };
  auto stageR = [&](int h, int buf, int kt) { if constexpr (WIDE) stageA(h, buf, kt); else stageB(h, buf, kt); };
  auto stageF = [&](int h, int buf, int kt) { if constexpr (WIDE) stageB(h, buf, kt); else stageA(h, buf, kt); };
  auto quadFR = [&](int fh, int rh) { if constexpr (WIDE) quadrant(rh, fh); else quadrant(fh, rh); };
#define AFX_BAR()                          \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

  // prologue of an output tile: all of K-tile 0 and three half-tiles of K-tile 1 go out
  auto issue_prologue = [&] {
    stageR(0, 0, 0);
    stageF(0, 0, 0);
    stageR(1, 0, 0);
    stageF(1, 0, 0);
    if (nk > 1) {
      stageR(0, 1, 1);
      stageF(0, 1, 1);
      stageR(1, 1, 1);
    }
  };

  auto ktile = [&](auto bufc, int t) {
    constexpr int b = decltype(bufc)::value;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // ---- phase 1
    readR(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    readF(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more1 && !(AFX_DBG(p, 128))) stageF(1, b ^ 1, t + 1);  // (AFX_DBG 128 / 256 / 512 / 1024, timing only: one operand half-tile's DMA off)
    // the R0 reads (issued first) are done -- all but the F0 reads behind them: R0 may be re-staged next phase
    if constexpr ((WIDE ? NTH : MF0) * KSN == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if constexpr ((WIDE ? NTH : MF0) * KSN == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AFX_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quadFR(0, 0);
    AFX_BAR();
    // ---- phase 2
    readR(b, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2 && !(AFX_DBG(p, 256))) stageR(0, b, t + 2);
    AFX_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quadFR(0, 1);
    AFX_BAR();
    // ---- phase 3
    readF(b, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2 && !(AFX_DBG(p, 512))) stageF(0, b, t + 2);
    AFX_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quadFR(1, 1);
    AFX_BAR();
    // ---- phase 4
    if (more2) {
      if (!(AFX_DBG(p, 1024))) stageR(1, b, t + 2);
      wait_vmcnt<6>();  // tile t+1 has landed; R0/F0/R1 of tile t+2 stay in flight
    } else {
      wait_vmcnt<0>();
    }
    AFX_BAR();
    quadFR(1, 0);
    AFX_BAR();
  };
  // ---- the K-tile as TWO phases of 32 MFMAs (PH2) ---------------------------------------------------------------
  // Measured with everything but the MFMAs and the barriers switched off (tools/bench_kloop_attr.py), the 4-phase
  // loop still takes 1.31 us per K-tile against 0.98 us of matrix-pipe time: every hand-over of the pipe from one
  // wave row to the other (s_barrier release, first issue) costs ~90 cycles, and a 16-MFMA segment is only 256.
  // Two phases per K-tile halve the hand-overs:
  //     phase A: read R0, R1, F0 (16 ds_read_b128)   stage F1(t+1)          MFMA (F0,R0) (F0,R1)
  //     phase B: read F1 (8)                         stage R0 R1 F0 (t+2)   MFMA (F1,R1) (F1,R0)
  // each phase = { reads ; LDS-DMA ; vmcnt ; lgkmcnt(0) ; s_barrier ; 32 MFMA ; s_barrier }; same registers as before
  // (both R halves were resident already).  Hazards, in barrier intervals of wave row 0 (row 1 one interval later):
  //   WAR  every read is RETIRED (lgkmcnt(0)) before its phase's first barrier, so a slot is free two intervals after
  //        row 0 read it: R0 R1 F0 of tile t (read in A_t) are re-staged from B_t on, F1 (read in B_t) from A_t+1 on.
  //   RAW  a wave waits for its own DMA at the END of a read interval, one full interval before the earliest reader
  //        of the other row needs it: end of B_t-1 -> R0 R1 F0 of tile t (read from A_t), end of A_t -> F1 of tile t
  //        (read from B_t).  In issue order the queue then holds [F1(t+1) 2] [R0R1F0(t+2) 6] behind what is needed:
  //        vmcnt(8) in steady state; every transfer has a whole K-tile of time to land.
  unsigned long long* ts_lds = (unsigned long long*)(smem + 2 * BUF);  // [8 waves][16 K-tiles][16 stamps]
  bool ts_on = TS;
  auto stamp = [&](int t, int slot) {
    if constexpr (TS) {
      if (ts_on && t < 16) {
        const unsigned long long c = __builtin_readcyclecounter();
        if (lane == 0) ts_lds[(wave * 16 + t) * 16 + slot] = c;
      }
    }
  };
  auto ktile2 = [&](auto bufc, int t) {
    constexpr int b = decltype(bufc)::value;
    constexpr int NF1 = WIDE ? DB : DA, NRRF = 2 * (WIDE ? DA : DB) + (WIDE ? DB : DA);  // DMAs per wave: F1; R0+R1+F0
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // ---- phase A
    stamp(t, 0);
    readR(b, 0);
    readR(b, 1);
    readF(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more1 && !(AFX_DBG(p, 128))) stageF(1, b ^ 1, t + 1);
    if (more1) wait_vmcnt<NF1 + NRRF>();  // F1(t) has landed; R0R1F0(t+1), F1(t+1) stay in flight
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(t, 1);
    AFX_BAR();
    stamp(t, 2);
    quadFR(0, 0);
    quadFR(0, 1);
    stamp(t, 3);
    AFX_BAR();
    stamp(t, 4);
    // ---- phase B
    readF(b, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      if (!(AFX_DBG(p, 256))) stageR(0, b, t + 2);
      if (!(AFX_DBG(p, 1024))) stageR(1, b, t + 2);
      if (!(AFX_DBG(p, 512))) stageF(0, b, t + 2);
      wait_vmcnt<NF1 + NRRF>();  // R0R1F0(t+1) have landed; F1(t+1), R0R1F0(t+2) stay in flight
    } else if (more1) {
      wait_vmcnt<NF1>();  // F1(t+1) may stay in flight
    } else {
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(t, 5);
    AFX_BAR();
    stamp(t, 6);
    quadFR(1, 1);
    quadFR(1, 0);
    stamp(t, 7);
    AFX_BAR();
    stamp(t, 8);
  };
  auto ktile_r3 = [&](auto rbc, int t, int fb, int fb2) {  // fb / fb2: byte offsets of A's ring slots of tiles t / t+2
    constexpr int rb = decltype(rbc)::value;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // ---- phase A
    readB3(rb, 0);
    readB3(rb, 1);
    readA3(fb, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      stageA3(0, fb2, t + 2);
      stageA3(1, fb2, t + 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AFX_BAR();
    quadrant(0, 0);
    quadrant(0, 1);
    AFX_BAR();
    // ---- phase B
    readA3(fb, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      stageB3(0, rb, t + 2);
      stageB3(1, rb, t + 2);
      wait_vmcnt<2 * DA + 2 * DB>();  // tile t+1 (issued one K-tile and more ago) has landed; tile t+2 stays in flight
    } else {
      wait_vmcnt<0>();
    }
    (void)more1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AFX_BAR();
    quadrant(1, 1);
    quadrant(1, 0);
    AFX_BAR();
  };
  auto issue_prologue_r3 = [&] {  // tiles 0 and 1 entirely
    stageB3(0, 0, 0);
    stageB3(1, 0, 0);
    stageA3(0, 0, 0);
    stageA3(1, 0, 0);
    if (nk > 1) {
      stageB3(0, 1, 1);
      stageB3(1, 1, 1);
      stageA3(0, FSZ3, 1);
      stageA3(1, FSZ3, 1);
    }
  };
  auto issue_prologue2 = [&] {
    stageR(0, 0, 0);
    stageR(1, 0, 0);
    stageF(0, 0, 0);
    stageF(1, 0, 0);
    if (nk > 1) {
      stageR(0, 1, 1);
      stageR(1, 1, 1);
      stageF(0, 1, 1);
    }
  };
  int v = blockIdx.x;
  setup(v);
  if constexpr (PH == 3) issue_prologue_r3(); else if constexpr (PH != 0) issue_prologue2(); else issue_prologue();
  bool first = true;
  for (;;) {
    // K-tile 0 has landed.  For the first output tile that is the counted wait of the template;
    // for later ones the epilogue stores of the previous tile were issued BEHIND these DMAs and
    // vmcnt retires in order, so the wait is vmcnt(0) (the stores were issued all through the
    // epilogue and are mostly acknowledged by now).
    if constexpr (PH == 3) {
      if (first && nk > 1) wait_vmcnt<2 * DA + 2 * DB>();  // K-tile 0 has landed, K-tile 1 may be in flight
      else wait_vmcnt<0>();
    } else if constexpr (PH != 0) {
      constexpr int NF1 = WIDE ? DB : DA, NRRF = 2 * (WIDE ? DA : DB) + (WIDE ? DB : DA);
      if (first && nk > 1) wait_vmcnt<NF1 + NRRF>();  // R0 R1 F0 of K-tile 0 have landed
      else if (first) wait_vmcnt<NF1>();
      else wait_vmcnt<0>();
    } else {
      if (first && nk > 1) wait_vmcnt<6>();
      else wait_vmcnt<0>();
    }
    first = false;
    AFX_BAR();
    if (wr == 1) AFX_BAR();  // the second wave row runs one barrier behind the first
#pragma unroll
    for (int i = 0; i < MF0 + MF1; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NTH; ++j) acc[i][j] = 0.f;
    // A wait the COMPILER can see: its waitcnt pass does not model the inline-asm waits above, so at the K-loop header
    // it still believed the loads of the previous output tile's epilogue (residual rows, a spill reload) outstanding
    // and protected the loop's first register writes with an s_waitcnt vmcnt(0) of its own -- at the header, i.e.
    // in EVERY iteration, draining the operand DMA queue every second K-tile.  Those loads are long complete here
    // (vmcnt retires in order and the asm waits above covered them); this tells the pass so.  Hardware cost: the first
    // output tile of a workgroup waits for all of its prologue instead of the first K-tile only.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only (gfx9 encoding: expcnt / lgkmcnt fields at their maximum)
    for (int t = 0; t < nk; t += 2) {
      if constexpr (PH == 3) {
        // A's ring slot of tile t is (t % 3) * FSZ3; t advances by 2 per iteration
        const int f0 = (t % 3) * FSZ3, f1 = ((t + 1) % 3) * FSZ3, f2 = ((t + 2) % 3) * FSZ3;
        ktile_r3(std::integral_constant<int, 0>{}, t, f0, f2);
        if (t + 1 < nk) ktile_r3(std::integral_constant<int, 1>{}, t + 1, f1, f0);
      } else if constexpr (PH == 1) {
        ktile2(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) ktile2(std::integral_constant<int, 1>{}, t + 1);
      } else {
        ktile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) ktile(std::integral_constant<int, 1>{}, t + 1);
      }
    }
    if (wr == 0) AFX_BAR();  // both wave rows are done with every LDS read of this output tile
    // The operand DMA of the NEXT output tile goes out before this tile's epilogue: its latency
    // and the first K-tile's fill hide behind the bias / activation / store work.  (The row-
    // LayerNorm epilogue keeps its scratch in the one half-tile the prologue does not write.)
    const int m0c = m0, n0c = n0;
    const int vn = v + gridDim.x;
    if (vn < nwg) {
      setup(vn);
      if constexpr (PH == 3) issue_prologue_r3(); else if constexpr (PH != 0) issue_prologue2(); else issue_prologue();
    }
    if (AFX_DBG(p, 64)) {  // timing only: no epilogue at all
#pragma unroll
      for (int i = 0; i < MF0 + MF1; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NTH; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[i][j][r]));
    } else {
      gemm_epilogue<HT, BMC, BN, 2, 4, ROWLN, true, S3>(p, acc, smem + BUF + (WIDE ? OFF_B1 : OFF_A1), m0c, n0c, g);
    }
    ts_on = false;
    if (vn >= nwg) break;
    v = vn;
  }
  if constexpr (TS) {
    __syncthreads();
    if (blockIdx.x == 0 && p.out_h)
      for (int i = tid; i < 8 * 16 * 16; i += 512) ((unsigned long long*)p.out_h)[i] = ts_lds[i];
  }
#undef AFX_BAR
}

template <class HT, int BM, int BN, bool ROWLN, int MF = BM / 32, int PH = 1, bool TS = false, bool S3 = false>
static hipError_t launch_gemm8_t(const GemmArgs& p, int groups, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 128 + (TS ? 16384 : 0) + (PH == 3 ? BM * 128 : 0);  // ring3: a third buffer for A
  static_assert(lds <= 160 * 1024, "two K-tile buffers must fit the 160 KB LDS");
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)gemm8_kernel<HT, BM, BN, ROWLN, MF, PH, TS, S3>, lds); e != hipSuccess) return e;
  static int n_cu_of[kMaxDevices] = {0};  // (benign if two threads fill the same slot: same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
  if (!n_cu_of[dev]) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorUnknown;
    cus &= ~7;  // a multiple of the 8 XCDs (the tile remap relies on it)
    n_cu_of[dev] = cus < 8 ? 8 : cus;
  }
  const int n_cu = n_cu_of[dev];
  const int tiles = ((p.N + BN - 1) / BN) * ((p.M - p.m_lo + MF * 32 - 1) / (MF * 32));
  dim3 grid(tiles < n_cu ? tiles : n_cu, 1, groups);  // persistent: at most one workgroup per CU
  hipLaunchKernelGGL((gemm8_kernel<HT, BM, BN, ROWLN, MF, PH, TS, S3>), grid, dim3(512), lds, s, p);
  return hipGetLastError();
}

// =======================================================================================
// gemm4_kernel: the 256x256 tile on FOUR waves -- one wave per SIMD, each owning 128 x 128 outputs, its 256 accumulator
// registers in the AGPR half of the 512-register budget a lone wave has (the one structural form of
// the 256-wide product that rounds 1-3 had not measured).  Against the 8-wave kernel above: a third fewer LDS fragment
// bytes per MFMA (32 ds_read_b128 per 256 MFMAs instead of 24 per 128), four barrier participants instead of eight, no
// hand-over of the matrix pipe between two waves of a SIMD -- ONE instruction stream carries MFMAs, fragment reads and
// LDS-DMA, with the fragments of the NEXT k-step requested under the MFMAs of this one.
//   LDS: two K-tile buffers of [A 256 rows | B 256 rows] x 128 B (the 8-wave kernel's row layout and swizzle), 128 KB.
//   K-tile t (buffer t & 1), fragments F(t, 0) already in registers:
//     request F(t, 1)                          (32 ds_read_b128 per wave and k-step: 8 A + 8 B fragments x 2)
//     64 MFMA on F(t, 0)
//     32 MFMA on F(t, 1), first half
//     vmcnt(0) + lgkmcnt(0) + s_barrier        tile t+1 has landed (issued a K-tile ago); everyone is done READING tile t
//     LDS-DMA of tile t+2 into tile t's buffer; request F(t+1, 0)
//     32 MFMA on F(t, 1), second half          (under which the new requests land)
//   One barrier per K-tile, placed inside the MFMA stream.  Same k order per output element as every other tile.
// Plain K only, fp16 / bf16 (no split-precision walk), 256-row tiles only.
// MEASURED AND NOT SHIPPED (profiles/r04_gemm4_ab.txt): bit-identical rows, 568 / 573 TFLOP/s on QKV / FC1 against the 8-wave
// kernel's 932 / 927, 1 176 against 1 389 at 8192^3 -- with two K-tile buffers at most one K-tile (64 KB per CU) is in flight,
// against the 8-wave ring's 96 KB, and a K-tile then costs 1.83 us where the matrix pipe needs 0.98.  The
// kernel exists in the attribution build only (make attr; tools/diag_gemm4.py, BENCH_SET=w4 tools/bench_gemm.py).
// =======================================================================================
#ifdef AFX_ATTR
template <class HT>
__global__ __launch_bounds__(256) void gemm4_kernel(GemmArgs p) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  constexpr int STAGE = 512 * 128;  // bytes: A rows 0..255, B rows 256..511
  constexpr int NP = 16;            // 1-KB LDS-DMA pieces per wave and K-tile (64 KB / 4 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = blockIdx.z;
  const T* Ag = (const T*)p.A + (long)g * p.g_a;
  const T* Wg = (const T*)p.W + (long)g * p.g_w;
  const int nN = (p.N + 255) / 256, nM = (p.M + 255) / 256;
  const int nwg = nM * nN;
  const int nk = p.K >> 6;
  const T* src[NP];
  int m0 = 0, n0 = 0;
  auto setup = [&](int v) {
    int pm, pn;
    int L = v;
    if (p.map_mode >= 1) {
      const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    if (p.map_mode == 2) {
      constexpr int GM = 8;
      const int width = GM * nN, grp = L / width, first = grp * GM;
      const int gsz = nM - first < GM ? nM - first : GM;
      pm = first + (L % width) % gsz;
      pn = (L % width) / gsz;
    } else {
      pm = L / nN;
      pn = L % nN;
    }
    m0 = pm * 256;
    n0 = pn * 256;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (i * 4 + wave) * 8 + (lane >> 3);   // row of the stage: < 256 A, else B
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      if (i < NP / 2) {  // pieces i * 4 + wave < 32: A rows
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        src[i] = Ag + (long)(m / p.rpb) * p.a_batch + (long)(m % p.rpb) * p.a_row + c * 8;
      } else {
        int n = n0 + row - 256;
        n = n < p.N ? n : p.N - 1;
        src[i] = Wg + (long)n * p.ldw + c * 8;
      }
    }
  };
  const unsigned lds_base = (unsigned)(size_t)smem;
  auto dma16 = [&](const T* s_, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(s_), "s"(dst) : "memory");
  };
  auto issue_tile = [&](int buf, int kt) {
    const long ko = (long)kt * 64;
#pragma unroll
    for (int i = 0; i < NP; ++i) dma16(src[i] + ko, buf * STAGE + (i * 4 + wave) * 1024);
  };
  const int frow = lane & 15, fsw = (frow >> 1) & 7, kq = lane >> 4;
  const char* aR = smem + (wr * 128 + frow) * 128;
  const char* bR = smem + (256 + wc * 128 + frow) * 128;
  auto read_frags = [&](int buf, int ks, V8 (&af)[8], V8 (&wf)[8]) {
    const int slot = ((ks * 4 + kq) ^ fsw) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = *(const V8*)(bR + buf * STAGE + j * (16 * 128) + slot);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *(const V8*)(aR + buf * STAGE + i * (16 * 128) + slot);
  };
  f32x4 acc[8][8];
  // One fragment read / one LDS-DMA piece, so that they can be slipped BETWEEN the MFMAs of the single instruction stream (an
  // MFMA holds the issue port for a few of its 16 cycles: a read or the five instructions of a DMA fit in the shadow).  Issued
  // as a block they cost their whole issue time as a matrix-pipe bubble (first version: 508 TFLOP/s on QKV against the 8-wave
  // kernel's 860 -- profiles/r04_gemm4_ab.txt).
  auto read_one = [&](int buf, int ks, int q, V8 (&af)[8], V8 (&wf)[8]) {  // q = 0..7 B fragments, 8..15 A fragments
    const int slot = ((ks * 4 + kq) ^ fsw) * 16;
    if (q < 8) wf[q] = *(const V8*)(bR + buf * STAGE + q * (16 * 128) + slot);
    else af[q - 8] = *(const V8*)(aR + buf * STAGE + (q - 8) * (16 * 128) + slot);
  };
  auto dma_one = [&](int buf, int kt, int i) { dma16(src[i] + (long)kt * 64, buf * STAGE + (i * 4 + wave) * 1024); };
#define AFX_BAR4()                         \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
  int v = blockIdx.x;
  setup(v);
  issue_tile(0, 0);
  if (nk > 1) issue_tile(1, 1);
  bool first = true;
  for (;;) {
    if (first && nk > 1) wait_vmcnt<NP>();  // K-tile 0 has landed, K-tile 1 may be in flight
    else wait_vmcnt<0>();                   // (later output tiles: the epilogue's stores sit behind these DMAs in the queue)
    first = false;
    AFX_BAR4();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) the compiler can see (as in gemm8_kernel: keeps its own vmcnt(0) out of the loop)
    V8 fa0[8], fw0[8], fa1[8], fw1[8];
    read_frags(0, 0, fa0, fw0);
    for (int t = 0; t < nk; ++t) {
      const int b = t & 1;
      const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
      // ---- k-step 0: 64 MFMAs on F(t, 0); the 16 reads of F(t, 1) ride behind the first 16
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[i][j] = HT::mfma(fw0[j], fa0[i], acc[i][j]);
          if (i < 2) {
            read_one(b, 1, i * 8 + j, fa1, fw1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      // ---- k-step 1, first half: 32 MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = HT::mfma(fw1[j], fa1[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // every fragment of tile t is in registers on EVERY path from here on (a wait the compiler sees: without it the
      // join behind the conditional below made the second half wait for the NEW reads -- lgkmcnt(3..0) -- on the path that took it)
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
      if (more1) {
        wait_vmcnt<0>();  // tile t+1 (this wave's pieces) has landed: issued a whole K-tile ago
        AFX_BAR4();       // ... everyone's has; everyone is done reading tile t
      }
      // ---- k-step 1, second half: 32 MFMAs; behind the first 16 the reads of F(t+1, 0), behind the last 16 the DMA of tile t+2
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 4; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[i][j] = HT::mfma(fw1[j], fa1[i], acc[i][j]);
          const int q = (i - 4) * 8 + j;
          if (q < 16) {
            if (more1) read_one(b ^ 1, 0, q, fa0, fw0);
          } else {
            if (more2) dma_one(b, t + 2, q - 16);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AFX_BAR4();  // every wave is done reading the last K-tile: the next output tile's DMA may overwrite both buffers
    const int m0c = m0, n0c = n0;
    const int vn = v + gridDim.x;
    if (vn < nwg) {
      setup(vn);
      issue_tile(0, 0);
      if (nk > 1) issue_tile(1, 1);
    }
    gemm_epilogue<HT, 256, 256, 2, 2, false, true, false>(p, acc, smem, m0c, n0c, g);
    if (vn >= nwg) break;
    v = vn;
  }
#undef AFX_BAR4
}

template <class HT>
static hipError_t launch_gemm4_t(const GemmArgs& p, int groups, hipStream_t s) {
  constexpr int lds = 2 * 512 * 128;
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)gemm4_kernel<HT>, lds); e != hipSuccess) return e;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorInvalidDevice;
  cus &= ~7;
  if (cus < 8) cus = 8;
  const int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256);
  dim3 grid(tiles < cus ? tiles : cus, 1, groups);
  hipLaunchKernelGGL((gemm4_kernel<HT>), grid, dim3(256), lds, s, p);
  return hipGetLastError();
}
// gemm5_kernel: the 4-wave 256x256 tile again, this time built around BYTES IN FLIGHT.  gemm4_kernel above showed that the loop
// is paced by how many operand bytes a CU has outstanding (64 KB: 1.83 us per K-tile, the 8-wave ring's 96 KB: 1.57, matrix pipe:
// 0.98).  Here LDS is a ring of FIVE stages of one k-step each (32 k: [A 256 rows | B 256 rows] x 64 B = 32 KB), and a stage is
// only in use while its fragments travel to registers -- the free arch VGPRs of the AGPR-accumulator wave hold the current and
// the next k-step's fragments (128 registers) -- so three to four stages (96-128 KB per CU) are in flight at any time.
//   k-step t (64 MFMAs per wave on F(t), in registers):
//     vmcnt: stage t+1 has landed (stages t+2, t+3 stay in flight)   s_barrier: ... for every wave; stage t-1 is free
//     behind MFMAs 0..15: the 16 fragment reads of F(t+1)            behind MFMAs 16..23: the 8 LDS-DMA pieces of stage t+4 (into t-1's slot)
// 64-byte LDS rows: chunk c of row r at slot c ^ ((r >> 1) & 3) -- conflict-free for ds_read_b128's lane groups (brute-forced).
// Same k order per output element as every other tile.  Plain K, fp16 / bf16, 256-row tiles.  Attribution build (make attr).
template <class HT>
__global__ __launch_bounds__(256) void gemm5_kernel(GemmArgs p) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  constexpr int NST = 5, STAGE = 512 * 64, NP = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = blockIdx.z;
  const T* Ag = (const T*)p.A + (long)g * p.g_a;
  const T* Wg = (const T*)p.W + (long)g * p.g_w;
  const int nN = (p.N + 255) / 256, nM = (p.M + 255) / 256;
  const int nwg = nM * nN;
  const int nks = p.K >> 5;  // k-steps of 32
  const T* src[NP];
  int m0 = 0, n0 = 0;
  auto setup = [&](int v) {
    int pm, pn;
    int L = v;
    if (p.map_mode >= 1) {
      const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    if (p.map_mode == 2) {
      constexpr int GM = 8;
      const int width = GM * nN, grp = L / width, first = grp * GM;
      const int gsz = nM - first < GM ? nM - first : GM;
      pm = first + (L % width) % gsz;
      pn = (L % width) / gsz;
    } else {
      pm = L / nN;
      pn = L % nN;
    }
    m0 = pm * 256;
    n0 = pn * 256;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (i * 4 + wave) * 16 + (lane >> 2);  // row of the stage: < 256 A, else B (pieces i < 4 are A rows)
      const int c = (lane & 3) ^ ((row >> 1) & 3);
      if (i < NP / 2) {
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        src[i] = Ag + (long)(m / p.rpb) * p.a_batch + (long)(m % p.rpb) * p.a_row + c * 8;
      } else {
        int n = n0 + row - 256;
        n = n < p.N ? n : p.N - 1;
        src[i] = Wg + (long)n * p.ldw + c * 8;
      }
    }
  };
  const unsigned lds_base = (unsigned)(size_t)smem;
  auto dma16 = [&](const T* s_, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(s_), "s"(dst) : "memory");
  };
  auto dma_one = [&](int st, int ks, int i) { dma16(src[i] + (long)ks * 32, st * STAGE + (i * 4 + wave) * 1024); };
  auto issue_stage = [&](int st, int ks) {
#pragma unroll
    for (int i = 0; i < NP; ++i) dma_one(st, ks, i);
  };
  const int frow = lane & 15, kq = lane >> 4;
  const int slot = (kq ^ ((frow >> 1) & 3)) * 16;
  const char* aR = smem + (wr * 128 + frow) * 64 + slot;
  const char* bR = smem + (256 + wc * 128 + frow) * 64 + slot;
  auto read_one = [&](int st, int q, V8 (&af)[8], V8 (&wf)[8]) {  // q = 0..7 B fragments, 8..15 A fragments
    if (q < 8) wf[q] = *(const V8*)(bR + st * STAGE + q * (16 * 64));
    else af[q - 8] = *(const V8*)(aR + st * STAGE + (q - 8) * (16 * 64));
  };
  f32x4 acc[8][8];
#define AFX_BAR5()                         \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
  // one k-step: 64 MFMAs on (fa, fw); behind them the reads of the next k-step's fragments into (na, nw) and the DMA of k-step t+4.
  // MORE1 / MORE4 are compile-time (is there a k-step t+1 / t+4): the steady-state body carries no branch between its MFMAs
  // (with run-time tests the compiler turned every slipped-in read / DMA into a branch and shuffled fragments through AGPRs).
  auto kstep = [&](auto more1_c, auto more4_c, int behind, int t, int rd, int wrs, const V8 (&fa)[8], const V8 (&fw)[8], V8 (&na)[8],
                   V8 (&nw)[8]) {
    constexpr bool MORE1 = decltype(more1_c)::value, MORE4 = decltype(more4_c)::value;
    if constexpr (MORE1) {
      if constexpr (MORE4) wait_vmcnt<2 * NP>();  // stage t+1 has landed; t+2, t+3 stay in flight
      else if (behind >= 2) wait_vmcnt<2 * NP>();
      else if (behind == 1) wait_vmcnt<NP>();
      else wait_vmcnt<0>();
      AFX_BAR5();  // ... for every wave; every wave is done with stage t-1 (its fragments were consumed in k-step t-1)
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[i][j] = HT::mfma(fw[j], fa[i], acc[i][j]);
        const int q = i * 8 + j;
        if (q < 16) {
          if constexpr (MORE1) read_one(rd, q, na, nw);
          __builtin_amdgcn_sched_barrier(0);
        } else if (q < 16 + NP) {
          if constexpr (MORE4) dma_one(wrs, t + 4, q - 16);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  int v = blockIdx.x;
  setup(v);
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nks) issue_stage(st, st);
  bool first = true;
  for (;;) {
    if (first) {  // stage 0 has landed; up to three more stages in flight
      const int inflight = (nks < NST - 1 ? nks : NST - 1) - 1;
      if (inflight >= 3) wait_vmcnt<3 * NP>();
      else if (inflight == 2) wait_vmcnt<2 * NP>();
      else if (inflight == 1) wait_vmcnt<NP>();
      else wait_vmcnt<0>();
    } else {
      wait_vmcnt<0>();  // (later output tiles: the epilogue's stores sit behind the prologue's DMAs in the queue)
    }
    first = false;
    AFX_BAR5();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) the compiler can see (keeps its own vmcnt(0) out of the loop; as in gemm8_kernel)
    V8 fa0[8], fw0[8], fa1[8], fw1[8];
#pragma unroll
    for (int q = 0; q < 16; ++q) read_one(0, q, fa0, fw0);
    int rd = 1, wrs = NST - 1;  // slot of k-step t+1 (read next); slot of k-step t+4 (written next) = the slot of k-step t-1
    auto adv = [&] {
      rd = rd + 1 == NST ? 0 : rd + 1;
      wrs = wrs + 1 == NST ? 0 : wrs + 1;
    };
    const std::true_type Y{};
    const std::false_type N_{};
    int t = 0;
    // steady state, two k-steps per iteration (the two fragment sets swap roles): every k-step has a t+1 and a t+4
    for (; t + 5 < nks; t += 2) {
      kstep(Y, Y, 2, t, rd, wrs, fa0, fw0, fa1, fw1);
      adv();
      kstep(Y, Y, 2, t + 1, rd, wrs, fa1, fw1, fa0, fw0);
      adv();
    }
    // tail (nks is even: K % 64 == 0): the remaining k-steps in pairs, no more DMA once t + 4 >= nks
    for (; t < nks; t += 2) {
      if (t + 4 < nks) kstep(Y, Y, 2, t, rd, wrs, fa0, fw0, fa1, fw1);
      else kstep(Y, N_, nks - 2 - t, t, rd, wrs, fa0, fw0, fa1, fw1);
      adv();
      if (t + 2 < nks) {
        if (t + 5 < nks) kstep(Y, Y, 2, t + 1, rd, wrs, fa1, fw1, fa0, fw0);
        else kstep(Y, N_, nks - 3 - t, t + 1, rd, wrs, fa1, fw1, fa0, fw0);
      } else {
        kstep(N_, N_, 0, t + 1, rd, wrs, fa1, fw1, fa0, fw0);
      }
      adv();
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    AFX_BAR5();  // every wave is done reading: the next output tile's DMA may overwrite the ring
    const int m0c = m0, n0c = n0;
    const int vn = v + gridDim.x;
    if (vn < nwg) {
      setup(vn);
#pragma unroll
      for (int st = 0; st < NST - 1; ++st)
        if (st < nks) issue_stage(st, st);
    }
    gemm_epilogue<HT, 256, 256, 2, 2, false, true, false>(p, acc, smem, m0c, n0c, g);
    if (vn >= nwg) break;
    v = vn;
  }
#undef AFX_BAR5
}

template <class HT>
static hipError_t launch_gemm5_t(const GemmArgs& p, int groups, hipStream_t s) {
  constexpr int lds = 5 * 512 * 64;
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)gemm5_kernel<HT>, lds); e != hipSuccess) return e;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorInvalidDevice;
  cus &= ~7;
  if (cus < 8) cus = 8;
  const int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256);
  dim3 grid(tiles < cus ? tiles : cus, 1, groups);
  hipLaunchKernelGGL((gemm5_kernel<HT>), grid, dim3(256), lds, s, p);
  return hipGetLastError();
}
#endif  // AFX_ATTR

template <class HT, int BM, int BN, int WR, int WC, bool ROWLN = false, bool LEAN = false, bool S3 = false>
static hipError_t launch_gemm_t(const GemmArgs& p, int groups, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 128;
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)gemm_kernel<HT, BM, BN, WR, WC, ROWLN, LEAN, S3>, lds); e != hipSuccess) return e;
  dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, groups);
  hipLaunchKernelGGL((gemm_kernel<HT, BM, BN, WR, WC, ROWLN, LEAN, S3>), grid, dim3(64 * WR * WC), lds, s, p);
  return hipGetLastError();
}

// =======================================================================================
// "Deep" four- / eight-wave tile for products whose tile count leaves ONE workgroup per CU (the teacher's N = 1024 products at
// M = 16 x 199: 200 tiles of 128 x 128).  gemm_kernel above lives on two or three co-resident workgroups covering each other: a
// workgroup alone runs one serial chain per K-tile -- barrier, fragment reads, wait, MFMAs -- at 0.64 us, half of what its CU can
// take in (tools/diag_small_tile_balance.py).  Here the chain is shortened inside the wave instead:
//   * NS K-tile buffers, LDS-DMA through inline asm, counted vmcnt: NS - 1 tiles in flight, the wait at the top of a K-tile
//     is for a tile issued NS - 1 iterations ago;
//   * both k-steps' fragments are requested in one block before the first MFMA (sched_barrier pins the order; the compiler
//     counts lgkmcnt, so the second step's reads land under the first step's MFMAs).
// Same k order, same MFMA, same epilogue as every other tile: bit-identical rows.  Plain K only (no chunked K); S3 = the
// split-precision walk over [xh.wh | xl.wh | xh.wl] (dtype "fp16x3").
// =======================================================================================
template <class HT, int BM, int BN, int WR, int WC, int NS, bool S3 = false>
__global__ __launch_bounds__(64 * WR * WC) void gemm_deep_kernel(GemmArgs p) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  constexpr int NW = WR * WC;
  constexpr int WM = BM / WR, WN = BN / WC;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int AI = BM / (8 * NW), WI = BN / (8 * NW), PER = AI + WI;
  static_assert(AI >= 1 && WI >= 1 && NS >= 3 && NS <= 4, "tile / stage shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int g = blockIdx.z;
  int pm, pn;
  {
    const int nN = (p.N + BN - 1) / BN, nM = (p.M + BM - 1) / BM;
    const int nwg = nM * nN;
    int L = blockIdx.x;
    if (p.map_mode >= 1) {
      const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    if (p.map_mode == 2) {
      constexpr int GM = 8;
      const int width = GM * nN, grp = L / width, first = grp * GM;
      const int gsz = nM - first < GM ? nM - first : GM;
      pm = first + (L % width) % gsz;
      pn = (L % width) / gsz;
    } else {
      pm = L / nN;
      pn = L % nN;
    }
  }
  const int m0 = pm * BM, n0 = pn * BN;
  const T* Ag = (const T*)p.A + (long)g * p.g_a;
  const T* Wg = (const T*)p.W + (long)g * p.g_w;
  const T* a_src[AI];
  const T* w_src[WI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    a_src[i] = Ag + (long)(m / p.rpb) * p.a_batch + (long)(m % p.rpb) * p.a_row + c * 8;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < p.N ? n : p.N - 1;
    w_src[i] = Wg + (long)n * p.ldw + c * 8;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K >> 6;
  const unsigned lds_base = (unsigned)(size_t)smem;
  auto dma16 = [&](const T* src, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto stage = [&](int buf, int kt) {
    const long k0 = (long)kt << 6;  // (split precision: pair-form operands, K counts halfs -- the same walk)
    const unsigned base = (unsigned)(buf * STAGE);
#pragma unroll
    for (int i = 0; i < AI; ++i) dma16(a_src[i] + k0, base + (i * NW + wave) * 1024);
#pragma unroll
    for (int i = 0; i < WI; ++i) dma16(w_src[i] + k0, base + A_BYTES + (i * NW + wave) * 1024);
  };
  const int frow = lane & 15, fsw = (frow >> 1) & 7;
  const int a_off = (wr * WM + frow) * 128;
  const int w_off = A_BYTES + (wc * WN + frow) * 128;
  int slot[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) slot[ks] = ((ks * 4 + (lane >> 4)) ^ fsw) * 16;
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nk) stage(t, t);
  int rb = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int behind = nk - 1 - kt;  // tiles issued after tile kt that may stay in flight: min(NS - 2, behind)
    if (behind >= NS - 2) wait_vmcnt<(NS - 2) * PER>();
    else if (NS > 3 && behind == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
    __syncthreads();  // tile kt landed for every wave; every wave is done with tile kt - 1
    if (kt + NS - 1 < nk) stage(rb == 0 ? NS - 1 : rb - 1, kt + NS - 1);  // into the buffer tile kt - 1 was read from
    const char* sb = smem + rb * STAGE;
    rb = rb + 1 == NS ? 0 : rb + 1;
    V8 af[2][MT], wf[2][NT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[ks][j] = *(const V8*)(sb + w_off + j * 16 * 128 + slot[ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i) af[ks][i] = *(const V8*)(sb + a_off + i * 16 * 128 + slot[ks]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (S3) {
      s3_mfma<HT, MT, NT>(wf[0], wf[1], af[0], af[1], acc);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(wf[ks][j], af[ks][i], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  gemm_epilogue<HT, BM, BN, WR, WC, false, true, S3>(p, acc, smem, m0, n0, g);
}

template <class HT, int BM, int BN, int WR, int WC, int NS, bool S3 = false>
static hipError_t launch_gemm_deep_t(const GemmArgs& p, int groups, hipStream_t s) {
  constexpr int lds = NS * (BM + BN) * 128;
  static_assert(lds <= 160 * 1024, "K-tile buffers must fit the 160 KB LDS");
  static LdsLimit lim;
  if (hipError_t e = lim.ensure((const void*)gemm_deep_kernel<HT, BM, BN, WR, WC, NS, S3>, lds); e != hipSuccess) return e;
  dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, groups);
  hipLaunchKernelGGL((gemm_deep_kernel<HT, BM, BN, WR, WC, NS, S3>), grid, dim3(64 * WR * WC), lds, s, p);
  return hipGetLastError();
}

// tuning knobs for A/B runs (tools/bench_gemm.py); -1 = use the defaults below
static int g_map_override = -1;
static int g_tile_override = -1;  // 0: 128x128 / 4 waves, 1: 256x256 / 8 waves
void gemm_set_map_mode(int m) { g_map_override = m; }
void gemm_set_tile(int t) { g_tile_override = t; }
static int g_ant_override = -1;  // non-temporal A loads: -1 auto (row-complete tile only), 0, 1
void gemm_set_a_nt(int v) { g_ant_override = v; }
static int g_small_deep = 1;  // A/B knob: 1 (default) = the deep form of the 128x64 tile where at most two tiles fall on a CU
void gemm_set_small_deep(int v) { g_small_deep = v != 0; }
static int g_deep = -1;  // row-complete conv tile: 0 = 2-stage kernel, otherwise (default) the 8-phase kernel (A/B knob)
void gemm_set_deep(int v) { g_deep = v; }
#ifdef AFX_ATTR
static int g_nodma = 0;  // timing-only epilogue knob bits (GemmArgs::dbg_nodma; WRONG results when set)
bool gemm_set_nodma(int v) { g_nodma = v; return true; }
#else
static constexpr int g_nodma = 0;
bool gemm_set_nodma(int v) { return v == 0; }  // the product build has no such switch
#endif

// true: K is walked linearly (no chunked addressing)
static bool plain_k(const GemmArgs& p) { return p.kchunk == p.K; }

// Host-side shape contract; anything else is a programming error in the caller.
static const char* check_gemm(const GemmArgs& p, int groups) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return "gemm: empty problem";
  if (p.K % 64) return "gemm: K must be a multiple of 64 (pad the weights)";
  if (p.N % 4) return "gemm: N must be a multiple of 4";
  if (p.kchunk <= 0 || p.kchunk % 64) return "gemm: kchunk must be a positive multiple of 64";
  if (p.rpb <= 0) return "gemm: rows-per-batch must be positive";
  if (!p.out_f && !p.out_h) return "gemm: no output";
  if (p.oh_pairs && (!p.k1 || (p.N & 7) || (p.g_n & 7) || (p.ldo_h & 31) || !p.out_h))
    return "gemm: pair-form output goes with the wide-store split-precision epilogue (N % 8 == 0, row stride % 32 == 0)";
  if (p.k1) {
    if (p.k1 % 32 || p.K != 2 * p.k1 || !p.pre_scale || !(p.a_inv > 0.f) || (p.oh_pairs && !(p.oh_scale > 0.f)))
      return "gemm: split precision needs pair-form operands (K = 2 k1 halfs, k1 % 32 == 0), the column scales and the operand scales";
    if (((size_t)p.A & 15) || ((size_t)p.W & 15) || (p.a_row & 63) || (p.a_batch & 63) || (p.g_a & 63) || (p.ldw & 63) || (p.g_w & 63) || (p.kchunk_stride & 63))
      return "gemm: split precision: pair-form rows start on whole 32-element groups";
  }
  if (p.ln_gamma) {
    if (p.N != 512 || groups != 1) return "gemm: the fused LayerNorm epilogue needs N == 512 (row-complete tile)";
    if (!p.ln_beta || !p.bias || p.resid || p.alpha != 1.f) return "gemm: fused LayerNorm epilogue: bias + LN + act only";
  }
  return nullptr;
}

bool gemm_is_narrow(int N) { return N <= 64 || (N > 128 && N < 256 && N % 128 != 0); }

// Which tile instance serves a problem: 0 = 128x128, 1 = 128x64, 2 = 256x256 (2-stage),
// 3 = the row-complete 128x512 tile with the fused LayerNorm epilogue (2-stage),
// 7 = 8-phase 256x256, 8 = 8-phase row-complete 128x512.
// 256x256 tiles halve the operand bytes per FLOP (the per-CU L2->LDS rate is what bounds
// this kernel) but quantise badly at M = B*199: measured faster only for the conv layers
// (huge M, N = 512) and the K = 4096 FFN product (tools/bench_gemm.py, profiles/).
// Row split for the 8-phase kernel: with T tiles on C CUs (one workgroup each) the last partial round
// costs a whole tile time however few tiles it holds (M = 64 x 199 rows: FC1 800 tiles = 3 rounds +
// 32, QKV 600 = 2 rounds + 88).  For a plain GEMM the leading rows that fill whole rounds exactly go
// to the 8-phase kernel and the remaining rows to the 128x128 kernel (two workgroups per CU), when
// those fit ONE round of it.  Returns the number of leading rows (0: no split).
static int g_split = 1;  // A/B knob
void gemm_set_split(int v) { g_split = v; }
static int g_s3_small = 0;  // A/B knob (split precision, small-M products): 0 = the fp16 dispatch's choice (deep 128x64), 1 = 128x128 2-stage, 2 = 128x64 2-stage
void gemm_set_s3_small(int v) { g_s3_small = v; }
static int gemm_split_rows(const GemmArgs& p, int groups) {
  constexpr int kCUs = 256;
  if (!g_split || groups != 1 || p.ln_gamma || !plain_k(p) || p.rpb < p.M || gemm_is_narrow(p.N)) return 0;
  const long nN = (p.N + 255) / 256, nM = (p.M + 255) / 256, tiles = nN * nM;
  if (tiles <= kCUs || tiles % kCUs == 0) return 0;
  const long nM1 = (tiles / kCUs) * kCUs / nN;  // whole rounds, whole row tiles
  if (nM1 < 1 || nM1 >= nM) return 0;
  const long rest = p.M - nM1 * 256;
  const long rest128 = ((rest + 127) / 128) * ((p.N + 127) / 128);
  if (rest128 > 2 * kCUs) return 0;  // the remainder would itself need a second round
  if ((tiles % kCUs) * 2 > kCUs && nM1 * nN < (tiles / kCUs) * kCUs) return 0;  // last round is more than half full anyway
  return (int)(nM1 * 256);
}

// Height of the 8-phase 256-wide tile for a product that is NOT round-split: fragments per wave row (8 = 256 rows).
// Cost model: rounds x (MF + 5) -- fitted to tools/bench_teacher_gemm.py (M = 12736, N = 4096: 33.4 / 30.3 / 27.3 / 25.6 us
// per round at 8 / 7 / 6 / 5 fragments): the operand DMA of a K-tile does not shrink with the height, only the MFMAs do.
static int g_ph4 = 0;  // A/B knob: 0 = default (256-wide tiles: two-phase K-tile over a three-buffer A ring; conv tile: two-phase, two buffers), 1 = the 4-phase K-tile (16-MFMA segments) on the full-height tiles, 2 = the two-buffer two-phase form on the 256-wide tiles
void gemm_set_ph4(int v) { g_ph4 = v; }
static int g_conv_split = 1;  // A/B knob: remainder split of multi-round conv layers
void gemm_set_conv_split(int v) { g_conv_split = v; }
static int g_fit = 1;  // A/B knob: 0 = always 256 rows, 1 = fitted, 5..8 = forced
void gemm_set_fit(int v) { g_fit = v; }
static int gemm8_fit_mf(const GemmArgs& p, long* cost_out = nullptr) {
  constexpr long kCUs = 256;
  const long nN = (p.N + 255) / 256;
  auto cost_of = [&](int mf) { return ((nN * ((p.M + mf * 32 - 1) / (mf * 32)) + kCUs - 1) / kCUs) * (mf + 5); };
  int best = 8;
  if (g_fit >= 5 && g_fit <= 8) {
    best = g_fit;
  } else if (g_fit) {
    for (int mf = 7; mf >= 5; --mf)
      if (cost_of(mf) < cost_of(best)) best = mf;
  }
  if (cost_out) *cost_out = cost_of(best);
  return best;
}

// The same for the row-complete 128x512 tile (4 fragments per wave row = 128 rows; 3 = 96, 2 = 64): the conv layers of
// a small batch (B = 16: 200 / 100 / 50 / 25 / 13 / 7 tiles of 128 rows on 256 CUs) spread over more CUs.  The weight
// panel flows at 64 KB per K-tile whatever the height, so a tile's time shrinks less than its rows: cost MF + 6 (41.9 / 37.2 / 33.5 us at 128 / 96 / 64 rows, one round each).
static int gemm8_fit_rowln(const GemmArgs& p) {
  constexpr long kCUs = 256;
  if (!g_fit) return 4;
  if (g_fit >= 12 && g_fit <= 14) return g_fit - 10;  // forced (A/B): 12, 13, 14
  int best = 4;
  long best_cost = 0;
  for (int mf = 4; mf >= 2; --mf) {
    const long tiles = (p.M + mf * 32 - 1) / (mf * 32);
    const long cost = ((tiles + kCUs - 1) / kCUs) * (mf + 6);
    if (mf == 4 || cost < best_cost) {
      best = mf;
      best_cost = cost;
    }
  }
  return best;
}

int gemm_tile_of(const GemmArgs& p, int groups) {
  if (p.ln_gamma) return g_deep != 0 && plain_k(p) ? 8 : 3;
  if (gemm_is_narrow(p.N)) {
    // (narrow products with at most two 128x64 tiles per CU: the deep form of the tile, as below)
    const long t64 = (long)((p.M + 127) / 128) * ((p.N + 63) / 64);
    const bool lean = (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0;
    if (g_small_deep && !p.no_deep && g_tile_override < 0 && groups == 1 && t64 <= 512 && plain_k(p) && lean && p.m_lo == 0) return 92;
    return 1;
  }
  if (groups != 1) return 0;
  if (g_tile_override == 5) return 1;  // 128x64 / 4 waves (forced)
#ifdef AFX_ATTR
  if (g_tile_override >= 6 && g_tile_override <= 9) return plain_k(p) && !p.k1 ? 84 + g_tile_override : 0;  // deep tiles (90..93), A/B
#else
  if (g_tile_override == 8)  // the deep 128x64 tile, forced (tests) -- where its lean epilogue applies
    return plain_k(p) && (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0 && p.m_lo == 0 ? 92 : 0;
#endif
#ifdef AFX_ATTR
  if (g_tile_override == 4) return plain_k(p) && !p.k1 && (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0 ? 4 : 0;  // 4-wave 256x256 (A/B)
  if (g_tile_override == 10) return plain_k(p) && !p.k1 && (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0 ? 5 : 0;  // ... with the 5-stage ring
#endif
  if (g_tile_override == 3) return plain_k(p) ? 7 : 0;  // 8-phase 256x256
  if (g_tile_override >= 0) return g_tile_override == 1 ? 2 : 0;
  // Wave-quantisation model fitted to tools/bench_gemm.py (profiles/r01_gemm_tile_ab*.txt):
  // the 8-phase 256x256 kernel is ~15 % faster per FLOP at K = 1024 (25 % at K = 4096) but
  // runs one workgroup per CU, a 128x128 tile two; pick the better fill of the 256 CUs.
  const long b128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const long b256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  if (b128 < 384) {  // small batches: halve the tile to spread over the chip
    if (p.k1 && g_s3_small) return g_s3_small == 1 ? 0 : 1;
    // at most two 128x64 tiles per CU (b128 <= 256: the teacher's N = 1024 products at B <= 16): the deep form of that tile --
    // three K-tile buffers, counted vmcnt, both k-steps' fragments requested ahead of the MFMAs -- FC2 at M = 16 x 199 50.3 ->
    // 43.3 us, out-proj 19.4 -> 17.7; at M = 8 x 199 41.6 -> 31.8 (tools/diag_deep_tiles.py).  With three workgroups per CU
    // (b128 > 256) the two-buffer form, which fits three, stays ahead.  Same k order: bit-identical rows.
    const bool lean = (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0;
    if (g_small_deep && !p.no_deep && b128 <= 256 && plain_k(p) && lean && p.m_lo == 0) return 92;
    return 1;
  }
  if (gemm_split_rows(p, groups) > 0) return 7;  // whole rounds on the 8-phase kernel + a 128x128 remainder
  const double e128 = (double)b128 / (double)(((b128 + 511) / 512) * 512);
  const double e256 = 1.15 * (double)b256 / (double)(((b256 + 255) / 256) * 256);
  if (e256 <= e128) return 0;
  return plain_k(p) ? 7 : (p.k1 ? 0 : 2);  // the 8-phase kernel where its addressing applies (no chunked K)
}

template <class HT>
static hipError_t dispatch(const GemmArgs& p, int tile, int groups, hipStream_t s) {
  // lean epilogue (no activation or erf-GELU, N % 8 == 0: every GEMM of the two models' fast paths) where it applies
  const bool lean = (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0 && !(AFX_DBG(p, 16));
  switch (tile) {
    case 1: return lean ? launch_gemm_t<HT, 128, 64, 2, 2, false, true>(p, groups, s) : launch_gemm_t<HT, 128, 64, 2, 2>(p, groups, s);
    case 2: return launch_gemm_t<HT, 256, 256, 2, 4>(p, groups, s);
    case 3: return launch_gemm_t<HT, 128, 512, 2, 4, true>(p, groups, s);
#ifdef AFX_ATTR  // measured, not faster (profiles/r03_k_small_experiments.txt): only in the attribution build, for tools/diag_deep_tiles.py
    case 90: return launch_gemm_deep_t<HT, 128, 128, 2, 2, 4>(p, groups, s);
    case 91: return launch_gemm_deep_t<HT, 128, 128, 2, 4, 4>(p, groups, s);
    case 93: return launch_gemm_deep_t<HT, 128, 128, 2, 2, 3>(p, groups, s);
#endif
    case 92: return launch_gemm_deep_t<HT, 128, 64, 2, 2, 3>(p, groups, s);
#ifdef AFX_ATTR
    case 4: return launch_gemm4_t<HT>(p, groups, s);
    case 5: return launch_gemm5_t<HT>(p, groups, s);
#endif
    case 7:
      if (g_ph4 == 0) return launch_gemm8_t<HT, 256, 256, false, 8, 3>(p, groups, s);  // default: the three-buffer ring form
#ifdef AFX_ATTR
      if (p.dbg_nodma & 8192) return launch_gemm8_t<HT, 256, 256, false, 8, 1, true>(p, groups, s);  // clock stamps (TS)
#endif
      return g_ph4 == 1 ? launch_gemm8_t<HT, 256, 256, false, 8, 0>(p, groups, s) : launch_gemm8_t<HT, 256, 256, false>(p, groups, s);
    case 75: return g_ph4 == 0 ? launch_gemm8_t<HT, 256, 256, false, 5, 3>(p, groups, s) : launch_gemm8_t<HT, 256, 256, false, 5>(p, groups, s);
    case 76: return g_ph4 == 0 ? launch_gemm8_t<HT, 256, 256, false, 6, 3>(p, groups, s) : launch_gemm8_t<HT, 256, 256, false, 6>(p, groups, s);
    case 77: return g_ph4 == 0 ? launch_gemm8_t<HT, 256, 256, false, 7, 3>(p, groups, s) : launch_gemm8_t<HT, 256, 256, false, 7>(p, groups, s);
    case 8: return g_ph4 ? launch_gemm8_t<HT, 128, 512, true, 4, 0>(p, groups, s) : launch_gemm8_t<HT, 128, 512, true>(p, groups, s);
    case 82: return launch_gemm8_t<HT, 128, 512, true, 2>(p, groups, s);
    case 83: return launch_gemm8_t<HT, 128, 512, true, 3>(p, groups, s);
    default: return lean ? launch_gemm_t<HT, 128, 128, 2, 2, false, true>(p, groups, s) : launch_gemm_t<HT, 128, 128, 2, 2>(p, groups, s);
  }
}

// split precision (DT_FP16X3): the same tiles with the three-segment K walk and the scaled, fp32-writing epilogue
static hipError_t dispatch_s3(const GemmArgs& p, int tile, int groups, hipStream_t s) {
  typedef FP16 HT;
  const bool lean = (p.act == ACT_NONE || p.act == ACT_GELU) && (p.N & 7) == 0;
  switch (tile) {
    case 1: return lean ? launch_gemm_t<HT, 128, 64, 2, 2, false, true, true>(p, groups, s) : launch_gemm_t<HT, 128, 64, 2, 2, false, false, true>(p, groups, s);
    case 92: return launch_gemm_deep_t<HT, 128, 64, 2, 2, 3, true>(p, groups, s);
    case 7: return launch_gemm8_t<HT, 256, 256, false, 8, 3, false, true>(p, groups, s);
    case 75: return launch_gemm8_t<HT, 256, 256, false, 5, 3, false, true>(p, groups, s);
    case 76: return launch_gemm8_t<HT, 256, 256, false, 6, 3, false, true>(p, groups, s);
    case 77: return launch_gemm8_t<HT, 256, 256, false, 7, 3, false, true>(p, groups, s);
    case 3:
    case 8: return launch_gemm8_t<HT, 128, 512, true, 4, 1, false, true>(p, groups, s);
    case 82: return launch_gemm8_t<HT, 128, 512, true, 2, 1, false, true>(p, groups, s);
    case 83: return launch_gemm8_t<HT, 128, 512, true, 3, 1, false, true>(p, groups, s);
    default: return lean ? launch_gemm_t<HT, 128, 128, 2, 2, false, true, true>(p, groups, s) : launch_gemm_t<HT, 128, 128, 2, 2, false, false, true>(p, groups, s);
  }
}
#define AFX_DISPATCH_GEMM(p, tile) ((p).k1 ? dispatch_s3(p, tile, groups, s) : (dtype == DT_BF16 ? dispatch<BF16>(p, tile, groups, s) : dispatch<FP16>(p, tile, groups, s)))

const char* launch_gemm(const GemmArgs& p_in, int dtype, int groups, hipStream_t s) {
  if (dtype == DT_FP32) return launch_gemm_f32(p_in, groups, s);
  if ((dtype == DT_FP16X3) != (p_in.k1 != 0)) return "gemm: the split-precision fields (k1, planes, column scales) go with DT_FP16X3 and only with it";
  if (const char* e = check_gemm(p_in, groups)) return e;
  GemmArgs p = p_in;
  p.map_mode = g_map_override >= 0 ? g_map_override : 2;
  int tile = gemm_tile_of(p, groups);
  if ((tile == 7 || tile == 8) && ((p.act != ACT_NONE && p.act != ACT_GELU) || (p.N & 7)))
    tile = tile == 7 ? 0 : 3;  // the 8-phase kernels carry the lean epilogue: everything else stays on the 2-stage tiles
  p.a_nt = g_ant_override >= 0 ? g_ant_override : (tile == 3 ? 1 : 0);
  p.dbg_nodma = g_nodma;
  int m1 = tile == 7 && g_tile_override < 0 ? gemm_split_rows(p, groups) : 0;
  if (m1 > 0) {
    // round split against a fitted height on the whole problem, in the same units: whole rounds at full height plus
    // the remainder kernel (~10: measured 84.9 us split against 82.2 us as 3 rounds of 224-row tiles for QKV at M = 12736,
    // 123.7 against 121.1 as 4 rounds for FC1 -- profiles/r02_gemm_dma_spread_ab.txt)
    long fitted = 0;
    gemm8_fit_mf(p, &fitted);
    const long split = (((long)(m1 / 256) * ((p.N + 255) / 256) + 255) / 256) * 13 + 10;
    if (g_fit == 1 && fitted <= split) m1 = 0;
  }
  if (m1 > 0) {  // rows [0, m1) on the 8-phase kernel, rows [m1, M) on the 128x128 kernel
    const size_t hs = 2;
    GemmArgs a = p, b = p;
    a.M = m1; a.rpb = m1; a.o_batch_rows = m1; a.oh_batch_rows = m1;
    b.M = p.M - m1; b.rpb = b.M; b.o_batch_rows = b.M; b.oh_batch_rows = b.M;
    b.A = (const char*)p.A + (size_t)m1 * p.a_row * hs;
    if (p.resid) b.resid = p.resid + (size_t)m1 * p.ldr;
    if (p.out_f) b.out_f = p.out_f + (size_t)m1 * p.ldo_f;
    if (p.out_h) b.out_h = (char*)p.out_h + (size_t)m1 * p.ldo_h * (p.k1 ? 4 : hs);
    hipError_t err = AFX_DISPATCH_GEMM(a, 7);
    if (err == hipSuccess) err = AFX_DISPATCH_GEMM(b, 0);
    return err == hipSuccess ? nullptr : hipGetErrorString(err);
  }
  if (tile == 7) {
    const int mf = gemm8_fit_mf(p);
    if (mf < 8) tile = 70 + mf;
  } else if (tile == 8) {
    const int mf = gemm8_fit_rowln(p);
    if (mf < 4) tile = 80 + mf;
    // Remainder split of a multi-round conv layer: T tiles of 128 rows on 256 CUs cost ceil(T / 256) rounds however few tiles
    // the last one holds (B = 64: layer 1 3200 tiles = 12.5 rounds, layer 2 1600 = 6.25, layer 3 800 = 3.125).  When the
    // last round is at most 3/8 full, the whole rounds run at 128 rows and the remaining rows as 64-row tiles of the same
    // kernel in a second launch (rows [m_lo, M) -- same rows, bit for bit).  Worth 2-4 % of such a layer (209 -> 203 us at
    // 800 tiles; the persistent grid's tail is cheaper than a full round, so less than the tile count suggests).
    const long t128 = (p.M + 127) / 128, frac = t128 % 256;
    if (mf == 4 && g_fit == 1 && g_conv_split && t128 > 256 && frac != 0 && frac <= 96 && p.K >= 1536 && p.m_lo == 0) {
      GemmArgs a = p, b = p;
      a.M = (int)((t128 / 256) * 256 * 128);
      b.m_lo = a.M;
      hipError_t err = AFX_DISPATCH_GEMM(a, 8);
      if (err == hipSuccess) err = AFX_DISPATCH_GEMM(b, 82);
      return err == hipSuccess ? nullptr : hipGetErrorString(err);
    }
  }
  const hipError_t err = AFX_DISPATCH_GEMM(p, tile);
  return err == hipSuccess ? nullptr : hipGetErrorString(err);
}

}  // namespace afx
