// Operand-delivery probe: how fast can ONE CU pull L2-resident data, per path --
//   mode 0  global_load_lds_dwordx4 (LDS-DMA: no VGPR destination, what the GEMM kernels use)
//   mode 1  global_load_dwordx4 into VGPRs (consumed by an xor, nothing written to LDS)
//   mode 2  global_load_dwordx4 into VGPRs, then ds_write_b128 to LDS (the register-staged path)
// 256 workgroups x 8 waves (one per CU), each wave keeps DEPTH 1-KB pieces in flight and walks a 64-KB slice of a buffer
// every workgroup shares (L2 / L1 hits after the first pass), or its own 64-KB slice of a 16-MB buffer ("own").
// Reported: GB/s per CU and bytes per clock at 2.1 GHz.
// hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void pull(const char* src, long wg_stride, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  const unsigned lds_base = (unsigned)(size_t)smem;
  u32x4 acc = {0u, 0u, 0u, 0u};
  // piece p of the 64-KB slice: 1 KB; a wave walks pieces wave, wave + 8, ... (8 per pass), DEPTH of them in flight
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; g += DEPTH) {
      u32x4 r[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int piece = (g + d) * 8 + wave;
        const char* p = base + piece * 1024 + lane * 16;
        if constexpr (MODE == 0) {
          unsigned keep;
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + piece * 1024);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
        } else {
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(p) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (MODE != 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          asm volatile("" : "+v"(r[d]));
          if constexpr (MODE == 1) acc ^= r[d];
          else *(u32x4*)(smem + ((g + d) * 8 + wave) * 1024 + lane * 16) = r[d];
        }
      }
    }
  }
  if constexpr (MODE == 2) {
    __syncthreads();
    acc = *(u32x4*)(smem + tid * 16);
  }
  if constexpr (MODE == 0) {
    __syncthreads();
    acc = *(u32x4*)(smem + tid * 16);
  }
  sink[blockIdx.x * 512 + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, long stride, unsigned* sink) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)pull<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((pull<MODE, DEPTH>), dim3(256), dim3(512), 65536, 0, buf, stride, 20, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((pull<MODE, DEPTH>), dim3(256), dim3(512), 65536, 0, buf, stride, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_cu = 65536.0 * iters;
  const double gbs = bytes_cu / (ms * 1e-3) / 1e9;
  printf("%-28s depth %d %-6s %7.1f GB/s per CU  %5.1f B/clk  (%.2f us per 64 KB, chip %.1f TB/s)\n", name, DEPTH, stride ? "own" : "shared",
         gbs, gbs / 2.1, ms * 1e3 / iters, gbs * 256 / 1e3);
}

int main() {
  char* buf;
  unsigned* sink;
  hipMalloc(&buf, 256L * 65536);
  hipMemset(buf, 1, 256L * 65536);
  hipMalloc(&sink, 256 * 512 * 4);
  for (long stride : {0L, 65536L}) {
    run<0, 2>("LDS-DMA", buf, stride, sink);
    run<0, 4>("LDS-DMA", buf, stride, sink);
    run<0, 8>("LDS-DMA", buf, stride, sink);
    run<1, 2>("load -> VGPR", buf, stride, sink);
    run<1, 4>("load -> VGPR", buf, stride, sink);
    run<1, 8>("load -> VGPR", buf, stride, sink);
    run<2, 2>("load -> VGPR -> ds_write", buf, stride, sink);
    run<2, 4>("load -> VGPR -> ds_write", buf, stride, sink);
    run<2, 8>("load -> VGPR -> ds_write", buf, stride, sink);
  }
  hipDeviceSynchronize();
  return 0;
}
