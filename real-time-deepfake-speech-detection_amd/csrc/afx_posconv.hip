// Positional convolution of the wav2vec2 encoder, gfx950:
//   x += GELU( Conv1d(1024, 1024, k = 128, pad = 64, groups = 16)(x)[: T] + bias )
// As a GEMM with chunked K (one 64-wide chunk per tap, afx_gemm.hip) every tap re-reads its
// 128-frame window from L2: the kernel ran at the L2 -> LDS rate (15 TB/s aggregate, 0.31 ms at
// B = 64).  Here the window SLIDES instead: a workgroup owns one group (64 in / 64 out channels) of
// two utterances, stages their whole time-padded input (T + 127 frames x 64 channels, 45 KB each)
// in LDS once, and walks the 128 taps with the A fragments of tap j read at row offset +j of the same
// slab.  Only the weights stream (one 64 x 64 tile = 8 KB per tap, a 4-deep LDS ring filled by
// LDS-DMA, counted vmcnt(2): two taps in flight across the barrier).  L2 traffic drops from
// 3.3 GB to 0.6 GB per launch; the kernel becomes MFMA / LDS-read bound.
//   LDS: slab [2][<= 352 rows][128 B] + ring [4][64 rows][128 B] = 120 KB; rows are XOR-swizzled
//        (16-B chunk ^ (row >> 1) & 7) on the DMA source address and on the read, as in the GEMM.
//   8 waves (two per SIMD, so one's fragment reads hide under the other's MFMAs); a wave owns up to 4
//        of the 2 x ceil(T/16) row tiles and all 4 output-channel tiles: per tap 8 weight-fragment +
//        8 input-fragment ds_read_b128 feed 32 MFMAs.
//   Workgroup -> (group, utterance pair) keeps two groups per XCD, so an XCD's L2 holds the 2 MB of
//        weights its workgroups share.
#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

namespace {
constexpr int PC_C = 1024, PC_CPG = 64, PC_TAPS = 128, PC_RING = 4, PC_NW = 8, PC_MAXT = 4;  // waves; row tiles per wave
}

template <class HT>
__global__ __launch_bounds__(64 * PC_NW) void posconv_kernel(PosConvArgs p) {
  typedef typename HT::T T;
  typedef typename HT::V8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int rows = p.slab_rows;                 // multiple of 8, >= T + 127
  const int slab_bytes = rows * 128;
  char* ring = smem + 2 * slab_bytes;
  const unsigned lds_base = (unsigned)(size_t)smem;
  // workgroup -> (group, pair): XCD x (= id & 7) serves groups 2x and 2x + 1
  const int id = blockIdx.x, k2 = id >> 3;
  const int grp = (id & 7) * 2 + (k2 & 1), pair = k2 >> 1;
  const int u0 = pair * 2;
  const int nslot = u0 + 1 < p.B ? 2 : 1;
  const T* xp = (const T*)p.xpad;
  const T* wg = (const T*)p.W + (long)grp * PC_CPG * (PC_CPG * PC_TAPS);

  auto dma16 = [&](const T* src, unsigned off) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  // ---- the input slabs, once -----------------------------------------------------------
  const int last_row = p.T + PC_TAPS - 1;  // xpad has T + 128 rows per utterance
  for (int s = 0; s < nslot; ++s) {
    const T* xb = xp + (long)(u0 + s) * p.xpad_batch + grp * PC_CPG;
    for (int q = wave; q < rows / 8; q += PC_NW) {
      int row = q * 8 + (lane >> 3);
      // slab swizzle: 16-byte chunk c of row r sits at c ^ (((r >> 1) & 3) << 1).  The fragment rows of tap j start
      // at ANY row (abase + j); the K-style swizzle c ^ ((r >> 1) & 7) of the weight ring is conflict-free only for
      // 16-row-aligned fragments (measured: 0.25 bank-conflict cycles per LDS cycle here), this one at every offset
      // (bit 0 of the chunk stays the lane's kq bit, so the two kq values of a ds_read_b128 group never meet).
      const int c = (lane & 7) ^ (((row >> 1) & 3) << 1);
      row = row < last_row ? row : last_row;
      dma16(xb + (long)row * PC_C + c * 8, s * slab_bytes + q * 1024);
    }
  }
  // ---- weight ring: tap j = 64 output rows x 64 input channels, 8 pieces of 8 rows ------
  auto issue_tap = [&](int j) {
    static_assert(PC_NW == 8, "one 8-row piece of the 64-row weight tile per wave");
    const int row = wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    dma16(wg + (long)row * (PC_CPG * PC_TAPS) + j * PC_CPG + c * 8, 2 * slab_bytes + (j & (PC_RING - 1)) * 8192 + wave * 1024);
  };
  issue_tap(0);
  issue_tap(1);
  issue_tap(2);

  // ---- this wave's row tiles -------------------------------------------------------------
  const int nrt = (p.T + 15) >> 4, ntiles = nslot * nrt;
  int abase[PC_MAXT], aoff[PC_MAXT];  // the lane's row inside its slab at tap 0; byte offset of the slab
#pragma unroll
  for (int i = 0; i < PC_MAXT; ++i) {
    const int q = wave + PC_NW * i;
    const int slot = q >= nrt ? 1 : 0, rt = q - slot * nrt;
    abase[i] = rt * 16 + l15;
    aoff[i] = slot * slab_bytes;
  }
  f32x4 acc[PC_MAXT][4];
#pragma unroll
  for (int i = 0; i < PC_MAXT; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wsw = (l15 >> 1) & 7;
  const int woff[2] = {l15 * 128 + ((kq ^ wsw) * 16), l15 * 128 + (((4 + kq) ^ wsw) * 16)};

  for (int j = 0; j < PC_TAPS; ++j) {
    // tap j (and, at j = 0, the slabs) has landed on every wave; taps j+1, j+2 stay in flight
    if (j + 2 < PC_TAPS) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (j + 1 < PC_TAPS) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (j + 3 < PC_TAPS) issue_tap(j + 3);  // into the slot of tap j-1: every wave is past it
    const char* wb = ring + (j & (PC_RING - 1)) * 8192;
    V8 wf[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[c][ks] = *(const V8*)(wb + c * 2048 + woff[ks]);
#pragma unroll
    for (int i = 0; i < PC_MAXT; ++i) {
      if (wave + PC_NW * i < ntiles) {  // wave-uniform
        const int R = abase[i] + j;  // slab-local row: the swizzle is the one its DMA used
        const int sw = ((R >> 1) & 3) << 1;
        const V8 a0 = *(const V8*)(smem + aoff[i] + R * 128 + ((kq ^ sw) * 16));
        const V8 a1 = *(const V8*)(smem + aoff[i] + R * 128 + (((4 + kq) ^ sw) * 16));
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = HT::mfma(wf[c][0], a0, acc[i][c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = HT::mfma(wf[c][1], a1, acc[i][c]);
      }
    }
  }

  // ---- epilogue: x += GELU(acc + bias), fp32 residual stream -------------------------------
#pragma unroll
  for (int i = 0; i < PC_MAXT; ++i) {
    const int q = wave + PC_NW * i;
    if (q >= ntiles) continue;
    const int slot = q >= nrt ? 1 : 0, rt = q - slot * nrt;
    const int t = rt * 16 + l15;
    if (t >= p.T) continue;
    float* xr = p.x + ((long)(u0 + slot) * p.T + t) * PC_C + grp * PC_CPG;
    f32x4 bv[4], rv[4];  // every load of the tile before its first store (shared vmcnt: afx_gemm.hip epilogue note)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 16 + kq * 4;
      bv[c] = *(const f32x4*)(p.bias + grp * PC_CPG + col);
      rv[c] = *(const f32x4*)(xr + col);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 16 + kq * 4;
      *(f32x4*)(xr + col) = rv[c] + gelu_poly4(acc[i][c] + bv[c]);
    }
  }
}

const char* launch_posconv(const PosConvArgs& p_in, int dtype, hipStream_t s) {
  PosConvArgs p = p_in;
  if (dtype == DT_FP32) return "posconv: half-precision operands only (exact mode uses the fp32 GEMM)";
  if (p.B <= 0 || p.T <= 0 || 2 * ((p.T + 15) / 16) > PC_NW * PC_MAXT || p.T > 224) return "posconv: 1..224 frames";
  p.slab_rows = (p.T + PC_TAPS - 1 + 7) & ~7;
  const int lds = 2 * p.slab_rows * 128 + PC_RING * 8192;
  if (lds > 160 * 1024) return "posconv: slab does not fit the LDS";
  static LdsLimit lim[2];
  hipError_t e = hipSuccess;
  const int npair = (p.B + 1) / 2;
  dim3 grid(16 * npair);
  if (dtype == DT_BF16) {
    e = lim[0].ensure((const void*)posconv_kernel<BF16>, lds);
    if (e == hipSuccess) hipLaunchKernelGGL(posconv_kernel<BF16>, grid, dim3(64 * PC_NW), lds, s, p);
  } else {
    e = lim[1].ensure((const void*)posconv_kernel<FP16>, lds);
    if (e == hipSuccess) hipLaunchKernelGGL(posconv_kernel<FP16>, grid, dim3(64 * PC_NW), lds, s, p);
  }
  if (e == hipSuccess) e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace afx
