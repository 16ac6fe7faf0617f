// Write-bandwidth microbenchmark: how the shape of a wave's store instruction changes the rate
// at which a GEMM epilogue can drain.  hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o /tmp/store_bw
//   mode 0: each wave-instruction writes 1 KB contiguous (64 lanes x 16 B)
//   mode 1: 16 rows x 64 B per instruction (the wide-store epilogue: 4 lanes x 16 B per row), the other 64 B of
//           each 128-B line written by the NEXT instruction of the same wave
//   mode 2: 16 rows x 64 B, the second half-line written much later (other half of the tile)
//   mode 3: 8 rows x 128 B per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(char* out, long row_bytes, int rows_per_wg) {
  // a workgroup writes a tile of rows_per_wg rows x 512 B (256 fp16 columns); 8 waves: wave w -> 128-B column slab (w&3), row half (w>>2)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  char* base = out + (long)blockIdx.x * rows_per_wg * row_bytes + (long)blockIdx.y * 512;
  const int r0 = (wave >> 2) * (rows_per_wg / 2);
  char* slab = base + (long)r0 * row_bytes + (wave & 3) * 128;
  const int nrow = rows_per_wg / 2;
  if (MODE == 0) {  // reinterpret the slab as contiguous 1-KB pieces of the whole tile (different addressing, same bytes)
    char* t = out + ((long)blockIdx.x * gridDim.y + blockIdx.y) * rows_per_wg * 512 + wave * (rows_per_wg * 64);
    for (int i = 0; i < rows_per_wg * 64 / 1024; ++i) *(u32x4*)(t + i * 1024 + lane * 16) = v;
  } else if (MODE == 1) {
    for (int r = 0; r < nrow; r += 16)
      for (int half = 0; half < 2; ++half)
        *(u32x4*)(slab + (long)(r + (lane & 15)) * row_bytes + half * 64 + (lane >> 4) * 16) = v;
  } else if (MODE == 2) {
    for (int half = 0; half < 2; ++half)
      for (int r = 0; r < nrow; r += 16)
        *(u32x4*)(slab + (long)(r + (lane & 15)) * row_bytes + half * 64 + (lane >> 4) * 16) = v;
  } else {
    for (int r = 0; r < nrow; r += 8) *(u32x4*)(slab + (long)(r + (lane & 7)) * row_bytes + (lane >> 3) * 16) = v;
  }
}

int main(int argc, char** argv) {
  const int M = (argc > 1 ? atoi(argv[1]) : 12736) / 256 * 256, N = 4096;  // fp16 output of the fc1 shape, whole tiles
  const long row_bytes = (long)N * 2, bytes = (long)M * row_bytes;
  char* d;
  hipMalloc(&d, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  dim3 grid(M / 256, N / 256);
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      for (int i = 0; i < 10; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(512), 0, 0, d, row_bytes, 256);
        if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(512), 0, 0, d, row_bytes, 256);
        if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(512), 0, 0, d, row_bytes, 256);
        if (mode == 3) hipLaunchKernelGGL(k<3>, grid, dim3(512), 0, 0, d, row_bytes, 256);
      }
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (rep == 2) printf("mode %d: %.1f us per launch, %.2f TB/s (%ld MB)\n", mode, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12, bytes >> 20);
    }
  }
  return 0;
}
