// Peak probes for this box ("nominal peaks must be re-measured on the box"):
//   1. dense fp16 MFMA rate: every wave issues v_mfma_f32_16x16x32_f16 back to back on 8 independent
//      accumulators, operands in registers (pseudo-random and all-zero operand values: the clock the chip
//      holds under matrix load depends on the data);
//   2. fp32 MFMA rate (v_mfma_f32_16x16x4_f32), the exact-mode instruction;
//   3. HBM: float4 copy (read + write), read-only sum, write-only fill, 2 GB.
// hipcc --offload-arch=gfx950 -O3 tools/peak_probe.hip -o /tmp/peak_probe && /tmp/peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// The loop body is ONE asm statement of 8 MFMAs on 8 accumulators held in place ("+v"): left to hipcc the
// eight f32x4 accumulators are shuffled through v_accvgpr_read / v_accvgpr_write every iteration (round 1's
// probe: 8 MFMA + 52 accvgpr moves + 3 s_nop per iteration, which measured the moves, not the instruction
// -- its "1.32 PFLOP/s ceiling of 16x16x32" was an artefact).  Check with
//   /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading ... : the loop must be 8 v_mfma + s_add + s_cmp + s_cbranch.
template <bool ZERO>
__global__ __launch_bounds__(256) void mfma_f16(float* out, int iters) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = ZERO ? (_Float16)0.f : (_Float16)(((l * 37 + i * 11) % 61 - 30) * 0.03125f);
    b[i] = ZERO ? (_Float16)0.f : (_Float16)(((l * 53 + i * 7) % 59 - 29) * 0.03125f);
  }
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n\t"
        "v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n\t"
        "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n\t"
        "v_mfma_f32_16x16x32_f16 %7, %8, %9, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a), "v"(b));
  }
  const f32x4 t = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
  out[blockIdx.x * 256 + l] = t[0] + t[1] + t[2] + t[3];
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <bool ZERO>
__global__ __launch_bounds__(256) void mfma_f16_32(float* out, int iters) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = ZERO ? (_Float16)0.f : (_Float16)(((l * 37 + i * 11) % 61 - 30) * 0.03125f);
    b[i] = ZERO ? (_Float16)0.f : (_Float16)(((l * 53 + i * 7) % 59 - 29) * 0.03125f);
  }
  f32x16 c[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) s += c[j][i];
  out[blockIdx.x * 256 + l] = s;
}
__global__ __launch_bounds__(256) void mfma_f32(float* out, int iters) {
  const int l = threadIdx.x;
  const float a = ((l * 37) % 61 - 30) * 0.03125f, b = ((l * 53) % 59 - 29) * 0.03125f;
  f32x4 c[8];
  for (int j = 0; j < 8; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  out[blockIdx.x * 256 + l] = s;
}
__global__ void copy4(const f32x4* in, f32x4* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void read4(const f32x4* in, float* out, size_t n) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
__global__ void fill4(f32x4* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

template <class F>
static float time_ms(F f, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  f();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  const int blocks = p.multiProcessorCount * 8, iters = 20000;  // 8 workgroups of 4 waves per CU
  const double fl16 = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 32, fl32 = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
  float ms = time_ms([&] { hipLaunchKernelGGL(mfma_f16<false>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
  printf("fp16 MFMA 16x16x32, pseudo-random operands: %7.1f TFLOP/s\n", fl16 / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(mfma_f16<true>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
  printf("fp16 MFMA 16x16x32, zero operands:          %7.1f TFLOP/s\n", fl16 / ms / 1e9);
  const double fl32x = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
  ms = time_ms([&] { hipLaunchKernelGGL(mfma_f16_32<false>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
  printf("fp16 MFMA 32x32x16, pseudo-random operands: %7.1f TFLOP/s\n", fl32x / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(mfma_f16_32<true>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
  printf("fp16 MFMA 32x32x16, zero operands:          %7.1f TFLOP/s\n", fl32x / ms / 1e9);
  // one wave per SIMD only (no inter-wave overlap): issue rate of a single dependent-free stream
  ms = time_ms([&] { hipLaunchKernelGGL(mfma_f16<false>, dim3(p.multiProcessorCount), dim3(256), 0, 0, out, iters); }, 3);
  printf("fp16 MFMA 16x16x32, ONE wave per SIMD:      %7.1f TFLOP/s\n", fl16 / 8 / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
  printf("fp32 MFMA 16x16x4:                          %7.1f TFLOP/s\n", fl32 / ms / 1e9);
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;
  f32x4 *x, *y;
  (void)hipMalloc(&x, bytes);
  (void)hipMalloc(&y, bytes);
  (void)hipMemset(x, 1, bytes);
  ms = time_ms([&] { hipLaunchKernelGGL(copy4, dim3(p.multiProcessorCount * 16), dim3(256), 0, 0, x, y, n); }, 5);
  printf("HBM copy  (2 GB read + 2 GB write):         %7.2f TB/s\n", 2.0 * bytes / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(read4, dim3(p.multiProcessorCount * 16), dim3(256), 0, 0, x, out, n); }, 5);
  printf("HBM read  (2 GB):                           %7.2f TB/s\n", 1.0 * bytes / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(fill4, dim3(p.multiProcessorCount * 16), dim3(256), 0, 0, y, n); }, 5);
  printf("HBM write (2 GB):                           %7.2f TB/s\n", 1.0 * bytes / ms / 1e9);
  return 0;
}
