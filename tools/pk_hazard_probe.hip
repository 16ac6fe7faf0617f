// Isolation attempt for the "two-stream step is not bit-stable" finding of round 4 (profiles/r04_two_stream_race.txt).
// What is established on the engine (tools/r04_race.sh): conv0_kernel<F32T>, compiled with its 10-tap loop in packed fp32 math, returns
// wrong LOW halves in lanes 48-63 when an fp16 GEMM kernel starts beside it; compiled scalar it does not.  The last writers of the
// elements that move have the form
//
//     v_pk_fma_f32 v[d:d+1], v[a:a+1], v[d:d+1], v[c:c+1] op_sel:[0,1,0]
//
// (the destination pair overwrites the source pair whose HIGH register the LOW result reads).  This probe asks whether that form is
// sufficient: forms A (above), B (the mirror: high half reads the LOW register, op_sel_hi:[1,0,1]), N (A's selects, destination distinct
// from every source) and S (the compiled kernel's own sequence: the pair fresh from LDS, three packed readers, the in-place writer last),
// each checked against scalar v_fma_f32 on the same inputs -- alone, beside a VALU kernel, beside a register-only fp16 MFMA kernel and
// beside an fp16 MFMA kernel streaming its operands from memory.  RESULT on MI355X (profiles/r04_pk_hazard_probe.txt): 0 wrong results
// in 3.9e8 wave-instructions in every case -- the form alone is NOT sufficient; what else the real pairing has (kernels starting and
// retiring beside the victim, the GEMMs' AGPR / LDS use, ...) is open.
//
//   hipcc --offload-arch=gfx950 -O2 tools/pk_hazard_probe.hip -o /tmp/pk_hazard_probe && /tmp/pk_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// counts[form][half][lane]
template <int FORM>
__global__ __launch_bounds__(256) void victim(unsigned* counts, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63;
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0x7FFF) * (1.0f / 32768.0f) - 0.5f; };
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 x = {rnd(), rnd()}, acc = {rnd(), rnd()};
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const f32x2 w = {rnd(), rnd()};
      // expected, with scalar instructions on copies (the asm barriers keep the compiler from packing or folding them)
      float xl = x[0], xh = x[1], wl = w[0], wh = w[1], al = acc[0], ah = acc[1];
      asm volatile("" : "+v"(xl), "+v"(xh), "+v"(wl), "+v"(wh), "+v"(al), "+v"(ah));
      float el, eh;
      f32x2 r;
      if (FORM == 0) {  // A: dst == src1, low result reads src1.hi  (lo = w.lo * x.hi + acc.lo, hi = w.hi * x.hi + acc.hi)
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(el) : "v"(wl), "v"(xh), "v"(al));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(eh) : "v"(wh), "v"(xh), "v"(ah));
        r = x;
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(r) : "v"(w), "v"(acc));
      } else if (FORM == 1) {  // B: dst == src1, high result reads src1.lo  (lo = w.lo * x.lo + acc.lo, hi = w.hi * x.lo + acc.hi)
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(el) : "v"(wl), "v"(xl), "v"(al));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(eh) : "v"(wh), "v"(xl), "v"(ah));
        r = x;
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(r) : "v"(w), "v"(acc));
      } else {  // N: the operand selects of A, destination distinct from every source
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(el) : "v"(wl), "v"(xh), "v"(al));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(eh) : "v"(wh), "v"(xh), "v"(ah));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(r) : "v"(w), "v"(x), "v"(acc));
      }
      asm volatile("" : "+v"(el), "+v"(eh));
      bad_lo += __float_as_uint(r[0]) != __float_as_uint(el);
      bad_hi += __float_as_uint(r[1]) != __float_as_uint(eh);
      // next step: the result becomes the accumulator, a fresh sample pair comes in (the shape of the conv tap loop)
      acc = f32x2{el, eh};
      x = f32x2{rnd(), rnd()};
    }
  }
  if (bad_lo) atomicAdd(counts + (FORM * 2 + 0) * 64 + lane, bad_lo);
  if (bad_hi) atomicAdd(counts + (FORM * 2 + 1) * 64 + lane, bad_hi);
}

// S: the sequence as hipcc emitted it in conv0_kernel<F32T> -- a sample pair fresh from LDS, three packed readers of its HIGH register
// (their own accumulators in place as source 2), then the in-place one that turns the sample pair into an accumulator.
__global__ __launch_bounds__(256) void victim_seq(unsigned* counts, int iters, unsigned seed) {
  __shared__ f32x2 xs[256 * 4];
  const int lane = threadIdx.x & 63;
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 9) & 0x7FFF) * (1.0f / 32768.0f) - 0.5f; };
  for (int i = 0; i < 4; ++i) xs[threadIdx.x * 4 + i] = f32x2{rnd(), rnd()};
  __syncthreads();
  unsigned bad_lo = 0, bad_hi = 0;
  f32x2 a0 = {rnd(), rnd()}, a1 = {rnd(), rnd()}, a2 = {rnd(), rnd()}, a3 = {rnd(), rnd()};
  for (int it = 0; it < iters; ++it) {
    const f32x2 w0 = {rnd(), rnd()}, w1 = {rnd(), rnd()}, w2 = {rnd(), rnd()}, w3 = {rnd(), rnd()};
    const unsigned addr = (unsigned)(size_t)(xs + threadIdx.x * 4 + (it & 3));
    // expected values: scalar fmas on the same inputs (the sample's high half through its own 32-bit LDS read)
    float xh;
    asm volatile("ds_read_b32 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(xh) : "v"(addr) : "memory");
    float e[8];
    const f32x2* ws[4] = {&w0, &w1, &w2, &w3};
    const f32x2* as[4] = {&a0, &a1, &a2, &a3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float wl = (*ws[k])[0], wh = (*ws[k])[1], al = (*as[k])[0], ah = (*as[k])[1];
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[2 * k]) : "v"(wl), "v"(xh), "v"(al));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[2 * k + 1]) : "v"(wh), "v"(xh), "v"(ah));
    }
    f32x2 x, r0 = a0, r1 = a1, r2 = a2;
    asm volatile(
        "ds_read_b64 %0, %4\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %1, %5, %0, %1 op_sel:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %6, %0, %2 op_sel:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %7, %0, %3 op_sel:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %8, %0, %9 op_sel:[0,1,0]"
        : "=&v"(x), "+v"(r0), "+v"(r1), "+v"(r2)
        : "v"(addr), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(a3)
        : "memory");
    const f32x2 got[4] = {r0, r1, r2, x};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned bl = __float_as_uint(got[k][0]) != __float_as_uint(e[2 * k]), bh = __float_as_uint(got[k][1]) != __float_as_uint(e[2 * k + 1]);
      if (k == 3) {  // the in-place instruction's result
        bad_lo += bl;
        bad_hi += bh;
      } else if (bl | bh) {
        atomicAdd(counts + 4 * 64 + lane, 1u);  // (a reader ahead of it moved: counted apart)
      }
    }
    a0 = f32x2{e[0], e[1]} * 0.5f; a1 = f32x2{e[2], e[3]} * 0.5f; a2 = f32x2{e[4], e[5]} * 0.5f; a3 = f32x2{e[6], e[7]} * 0.5f;
  }
  if (bad_lo) atomicAdd(counts + 0 * 64 + lane, bad_lo);
  if (bad_hi) atomicAdd(counts + 1 * 64 + lane, bad_hi);
}

__global__ __launch_bounds__(256) void aggressor_mfma(float* sink, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (i + 1));
  }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void aggressor_mfma_mem(float* sink, const f16x8* src, int n8, int iters) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
  unsigned i = (blockIdx.x * 256u + threadIdx.x) % n8;
  for (int it = 0; it < iters; ++it) {
    const f16x8 a = src[i], b = src[(i + 4099u) % n8];
    i = (i + 65537u) % n8;
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
  }
  if (c0[0] + c1[1] == 12345.678f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void aggressor_valu(float* sink, int iters) {
  float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
  for (int it = 0; it < iters; ++it) {
    v0 = __builtin_fmaf(v0, 0.999f, 0.001f);
    v1 = __builtin_fmaf(v1, 0.998f, 0.002f);
    v2 = __builtin_fmaf(v2, 0.997f, 0.003f);
    v3 = __builtin_fmaf(v3, 0.996f, 0.004f);
  }
  if (v0 + v1 + v2 + v3 == 12345.678f) sink[0] = 1.f;
}

template <int FORM>
static int run_case(const char* what, int beside, unsigned* d_counts, float* sink, hipStream_t s0, hipStream_t s1) {
  CK(hipMemsetAsync(d_counts, 0, 6 * 64 * 4, s0));
  CK(hipStreamSynchronize(s0));
  const int victim_iters = 400, rounds = 6;
  for (int r = 0; r < rounds; ++r) {
    // the aggressor first: one workgroup per CU (a wave per SIMD) for ~2 ms, the victim's waves fill the same SIMDs beside it
    if (beside == 1) hipLaunchKernelGGL(aggressor_valu, dim3(512), dim3(256), 0, s1, sink, 400000);
    if (beside == 2) hipLaunchKernelGGL(aggressor_mfma, dim3(512), dim3(256), 0, s1, sink, 60000);
    hipLaunchKernelGGL((victim<FORM>), dim3(4096), dim3(256), 0, s0, d_counts, victim_iters, 1234u + r);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned> h(6 * 64);
  CK(hipMemcpy(h.data(), d_counts, 6 * 64 * 4, hipMemcpyDeviceToHost));
  const double n = (double)rounds * 4096 * 4 * victim_iters * 10;  // wave-instructions checked
  for (int half = 0; half < 2; ++half) {
    unsigned long tot = 0, row[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) {
      tot += h[(FORM * 2 + half) * 64 + l];
      row[l >> 4] += h[(FORM * 2 + half) * 64 + l];
    }
    printf("  form %s, %-28s %s half: %8lu wrong lane-results in %.2e wave-instructions  (lanes 0-15: %lu, 16-31: %lu, 32-47: %lu, 48-63: %lu)\n",
           FORM == 0 ? "A" : FORM == 1 ? "B" : "N", what, half ? "HIGH" : "LOW ", tot, n, row[0], row[1], row[2], row[3]);
  }
  return 0;
}

static int run_seq(const char* what, int beside, unsigned* d_counts, float* sink, const f16x8* src, int n8, hipStream_t s0, hipStream_t s1) {
  CK(hipMemsetAsync(d_counts, 0, 6 * 64 * 4, s0));
  CK(hipStreamSynchronize(s0));
  const int victim_iters = 4000, rounds = 6;
  for (int r = 0; r < rounds; ++r) {
    if (beside == 1) hipLaunchKernelGGL(aggressor_valu, dim3(512), dim3(256), 0, s1, sink, 400000);
    if (beside == 2) hipLaunchKernelGGL(aggressor_mfma, dim3(512), dim3(256), 0, s1, sink, 60000);
    if (beside == 3) hipLaunchKernelGGL(aggressor_mfma_mem, dim3(512), dim3(256), 0, s1, sink, src, n8, 20000);
    hipLaunchKernelGGL(victim_seq, dim3(4096), dim3(256), 0, s0, d_counts, victim_iters, 99u + r);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned> h(6 * 64);
  CK(hipMemcpy(h.data(), d_counts, 6 * 64 * 4, hipMemcpyDeviceToHost));
  const double n = (double)rounds * 4096 * 4 * victim_iters;
  const char* nm[3] = {"in-place result, LOW  half", "in-place result, HIGH half", "the three readers ahead"};
  const int at[3] = {0, 1, 4};
  for (int k = 0; k < 3; ++k) {
    unsigned long tot = 0, row[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) {
      tot += h[at[k] * 64 + l];
      row[l >> 4] += h[at[k] * 64 + l];
    }
    printf("  sequence S, %-32s %-27s: %8lu wrong lane-results in %.2e sequences  (lanes 0-15: %lu, 16-31: %lu, 32-47: %lu, 48-63: %lu)\n", what, nm[k], tot, n,
           row[0], row[1], row[2], row[3]);
  }
  return 0;
}

int main() {
  unsigned* d_counts;
  float* sink;
  CK(hipMalloc(&d_counts, 6 * 64 * 4));
  CK(hipMalloc(&sink, 64));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s (%s), %d CUs\n", p.name, p.gcnArchName, p.multiProcessorCount);
  printf("A: v_pk_fma_f32 D, W, D, ACC op_sel:[0,1,0]     (destination = source 1; LOW result reads source 1's HIGH register)\n");
  printf("B: v_pk_fma_f32 D, W, D, ACC op_sel_hi:[1,0,1]  (destination = source 1; HIGH result reads source 1's LOW register)\n");
  printf("N: v_pk_fma_f32 R, W, X, ACC op_sel:[0,1,0]     (destination distinct from every source)\n");
  const char* names[3] = {"alone", "VALU kernel beside", "fp16 MFMA kernel beside"};
  for (int beside = 0; beside < 3; ++beside) {
    if (run_case<0>(names[beside], beside, d_counts, sink, s0, s1)) return 1;
    if (run_case<1>(names[beside], beside, d_counts, sink, s0, s1)) return 1;
    if (run_case<2>(names[beside], beside, d_counts, sink, s0, s1)) return 1;
  }
  printf("S: ds_read_b64 X; 3 x v_pk_fma_f32 Ai, Wi, X, Ai op_sel:[0,1,0]; v_pk_fma_f32 X, W3, X, A3 op_sel:[0,1,0]   (the compiled kernel's own sequence)\n");
  f16x8* src;
  const int n8 = 1 << 22;  // 64 MB of fp16 operands
  CK(hipMalloc(&src, (size_t)n8 * 16));
  CK(hipMemset(src, 0x11, (size_t)n8 * 16));
  const char* names2[4] = {"alone", "VALU kernel beside", "fp16 MFMA (registers) beside", "fp16 MFMA (streaming) beside"};
  for (int beside = 0; beside < 4; ++beside)
    if (run_seq(names2[beside], beside, d_counts, sink, src, n8, s0, s1)) return 1;
  return 0;
}
