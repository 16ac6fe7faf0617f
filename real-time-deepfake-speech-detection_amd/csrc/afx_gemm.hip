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

namespace afx {

// WR x WC waves per workgroup; each wave owns a (BM/WR) x (BN/WC) block of the tile.
template <class HT, int BM, int BN, int WR, int WC>
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

  auto stage = [&](int buf, int kt) {
    const int k0 = kt << 6;
    const long ka = (long)(k0 / p.kchunk) * p.kchunk_stride + (k0 % p.kchunk);
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + ka),
                                       (__attribute__((address_space(3))) void*)(base + (i * NW + wave) * 1024),
                                       16, 0, 0);
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (i * NW + wave) * 1024),
                                       16, 0, 0);
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
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int slot = ((ks * 4 + (lane >> 4)) ^ fsw) * 16;
      V8 af[MT], wf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const V8*)(sb + a_off + i * 16 * 128 + slot);
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *(const V8*)(sb + w_off + j * 16 * 128 + slot);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(wf[j], af[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
  const int gcol = g * p.g_n;
  const float alpha = p.alpha;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wr * WM + i * 16 + (lane & 15);
    if (m >= p.M) continue;
    const long orow = (long)(m / p.rpb) * p.o_batch_rows + (m % p.rpb) + p.o_row_off;
    const long hrow = (long)(m / p.rpb) * p.oh_batch_rows + (m % p.rpb) + p.oh_row_off;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) {
        const f32x4 b = *(const f32x4*)(p.bias + gcol + n);
        v += b;
      }
      if (p.act != ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      }
      v *= alpha;
      if (p.resid) {
        const f32x4 r4 = *(const f32x4*)(p.resid + orow * p.ldr + gcol + n);
        v += r4;
      }
      if (p.out_f) *(f32x4*)(p.out_f + orow * p.ldo_f + gcol + n) = v;
      if (p.out_h) {
        V4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (T)v[r];
        *(V4*)((T*)p.out_h + hrow * p.ldo_h + gcol + n) = h;
      }
    }
  }
}

template <class HT, int BM, int BN, int WR, int WC>
static hipError_t launch_gemm_t(const GemmArgs& p, int groups, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<HT, BM, BN, WR, WC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, groups);
  hipLaunchKernelGGL((gemm_kernel<HT, BM, BN, WR, WC>), grid, dim3(64 * WR * WC), lds, s, p);
  return hipGetLastError();
}

// tuning knobs for A/B runs (tools/bench_gemm.py); -1 = use the defaults below
static int g_map_override = -1;
static int g_tile_override = -1;  // 0: 128x128 / 4 waves, 1: 256x256 / 8 waves
void gemm_set_map_mode(int m) { g_map_override = m; }
void gemm_set_tile(int t) { g_tile_override = t; }

// Host-side shape contract; anything else is a programming error in the caller.
static const char* check_gemm(const GemmArgs& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return "gemm: empty problem";
  if (p.K % 64) return "gemm: K must be a multiple of 64 (pad the weights)";
  if (p.N % 4) return "gemm: N must be a multiple of 4";
  if (p.kchunk <= 0 || p.kchunk % 64) return "gemm: kchunk must be a positive multiple of 64";
  if (p.rpb <= 0) return "gemm: rows-per-batch must be positive";
  if (!p.out_f && !p.out_h) return "gemm: no output";
  return nullptr;
}

bool gemm_is_narrow(int N) { return N <= 64 || (N > 128 && N < 256 && N % 128 != 0); }

// Which tile instance serves a problem: 0 = 128x128, 1 = 128x64, 2 = 256x256.
// 256x256 tiles halve the operand bytes per FLOP (the per-CU L2->LDS rate is what bounds
// this kernel) but quantise badly at M = B*199: measured faster only for the conv layers
// (huge M, N = 512) and the K = 4096 FFN product (tools/bench_gemm.py, profiles/).
int gemm_tile_of(const GemmArgs& p, int groups) {
  if (gemm_is_narrow(p.N)) return 1;
  const bool big_auto = p.N <= 1024 && (long)p.M * p.K >= 12736L * 2048;
  const bool big = groups == 1 && (g_tile_override >= 0 ? g_tile_override == 1 : big_auto);
  return big ? 2 : 0;
}

const char* launch_gemm(const GemmArgs& p_in, int dtype, int groups, hipStream_t s) {
  if (const char* e = check_gemm(p_in)) return e;
  GemmArgs p = p_in;
  p.map_mode = g_map_override >= 0 ? g_map_override : 2;
  hipError_t err;
  const int tile = gemm_tile_of(p, groups);
  const bool narrow = tile == 1, big = tile == 2;
  if (dtype == DT_BF16)
    err = narrow ? launch_gemm_t<BF16, 128, 64, 2, 2>(p, groups, s)
                 : big ? launch_gemm_t<BF16, 256, 256, 2, 4>(p, groups, s) : launch_gemm_t<BF16, 128, 128, 2, 2>(p, groups, s);
  else
    err = narrow ? launch_gemm_t<FP16, 128, 64, 2, 2>(p, groups, s)
                 : big ? launch_gemm_t<FP16, 256, 256, 2, 4>(p, groups, s) : launch_gemm_t<FP16, 128, 128, 2, 2>(p, groups, s);
  return err == hipSuccess ? nullptr : hipGetErrorString(err);
}

}  // namespace afx
