// fp32 "exact mode" GEMM for gfx950: the same GemmArgs contract as afx_gemm.hip
// (conv-as-GEMM row addressing, chunked K for the grouped positional conv, groups,
// bias / activation / alpha / residual epilogue, fp32 + "operand" outputs with the
// row remaps) but with fp32 operands on v_mfma_f32_16x16x4_f32, so that a whole
// forward can be run with no reduced-precision rounding anywhere.  It exists for
// parity work (DESIGN.md "Numerics": fp16 operand rounding moves the AASIST graph
// pooling's discrete top-k on a few utterances; this mode removes that), not for
// the headline rate: the fp32 matrix rate of the chip is 1/16 of the fp16 one.
//
// Tile: 128 rows x 128 (or 64) columns per 256-thread workgroup; each wave owns 32 rows x
// 128 (64) columns (2 x 8 (4) MFMA accumulators).  Operands are read straight from global memory
// as 16-byte rows (lane (r, kq) reads k = 4 kq .. 4 kq + 3 of row r): the K loop is 16
// wide, 10 loads feed 64 MFMAs, the panel re-reads across workgroups are L2 hits.
// MFMA operands are swapped (W as "A") so a lane ends up with 4 consecutive columns.
#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

template <int NT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int grp = blockIdx.z;
  const long m0 = (long)blockIdx.y * 128 + wave * 32;
  const int n0 = blockIdx.x * (16 * NT);
  if (m0 >= p.M) return;
  const float* A = (const float*)p.A + grp * p.g_a;
  const float* W = (const float*)p.W + grp * p.g_w;
  const float* arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    long m = m0 + mt * 16 + r;
    m = m < p.M ? m : p.M - 1;
    arow[mt] = A + (m / p.rpb) * p.a_batch + (m % p.rpb) * p.a_row + kq * 4;
  }
  const float* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = n0 + nt * 16 + r;
    n = n < p.N ? n : p.N - 1;
    wrow[nt] = W + (long)n * p.ldw + kq * 4;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // K walk, 16 at a time, software-pipelined one step ahead in registers (a grid of a few
  // hundred workgroups leaves ~1 wave per SIMD: nothing else hides the load latency)
  const int spc = p.kchunk / 16;  // steps per chunk
  const int steps = (p.K / p.kchunk) * spc;
  f32x4 a[2], b[NT], an[2], bn[NT];
  auto load = [&](f32x4* ar, f32x4* br, int step) {
    const int ch = step / spc, k0 = (step - ch * spc) * 16;
    const long ao = (long)ch * p.kchunk_stride + k0;
    const long wo = (long)ch * p.kchunk + k0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ar[mt] = *(const f32x4*)(arow[mt] + ao);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) br[nt] = *(const f32x4*)(wrow[nt] + wo);
  };
  load(a, b, 0);
  for (int st = 0; st < steps; ++st) {
    if (st + 1 < steps) load(an, bn, st + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt][s], a[mt][s], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a[mt] = an[mt];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
  }
  // lane holds out[m = m0 + 16 mt + (lane & 15)][n = n0 + 16 nt + 4 (lane >> 4) + 0..3]
  const int gn = grp * p.g_n;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long m = m0 + mt * 16 + r;
    if (m >= p.M) continue;
    const long bi = m / p.rpb, ri = m % p.rpb;
    const long orow = bi * p.o_batch_rows + ri + p.o_row_off;
    const long hrow = bi * p.oh_batch_rows + ri + p.oh_row_off;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + nt * 16 + kq * 4;
      if (n >= p.N) continue;  // N % 4 == 0: a lane's 4 columns are all in or all out
      const int c = gn + n;
      f32x4 v = acc[mt][nt];
      if (p.bias) v += *(const f32x4*)(p.bias + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], p.act) * p.alpha;
      if (p.resid) v += *(const f32x4*)(p.resid + orow * p.ldr + c);
      if (p.out_f) *(f32x4*)(p.out_f + orow * p.ldo_f + c) = v;
      if (p.out_h) *(f32x4*)((float*)p.out_h + hrow * p.ldo_h + c) = v;
    }
  }
}

const char* launch_gemm_f32(const GemmArgs& p, int groups, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return "gemm(fp32): empty problem";
  if (p.N % 4 || p.kchunk <= 0 || p.kchunk % 16 || p.K % p.kchunk) return "gemm(fp32): need N % 4 == 0, K chunks % 16 == 0";
  if (p.ln_gamma) return "gemm(fp32): the fused LayerNorm epilogue is a half-precision tile feature";
  if (!p.out_f && !p.out_h) return "gemm(fp32): no output";
  if (p.rpb <= 0) return "gemm(fp32): rows per batch must be positive";
  const long mt = (p.M + 127) / 128;
  // 128-column tiles unless that leaves the 256 CUs under two workgroups each
  if (p.N > 64 && mt * ((p.N + 127) / 128) * groups >= 512)
    hipLaunchKernelGGL(gemm_f32_kernel<8>, dim3((p.N + 127) / 128, (unsigned)mt, groups), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<4>, dim3((p.N + 63) / 64, (unsigned)mt, groups), dim3(256), 0, s, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace afx
