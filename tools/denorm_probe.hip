// Does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs, or flush them to zero?  (Split precision stores the low part of
// an activation as an fp16 value: for small activations that low part is subnormal, so the answer decides how much the
// activation scale of the hi / lo pairs matters -- DESIGN.md section 5.)
//   hipcc --offload-arch=gfx950 -O2 tools/denorm_probe.hip -o /tmp/denorm_probe && /tmp/denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(float* out, float aval, float bval) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)0.f;
    b[i] = (_Float16)0.f;
  }
  // every lane: k-element 0 of its row / column -> C[m][n] = sum over the 4 lane groups' k = 0 entries = 4 * a * b
  a[0] = (_Float16)aval;
  b[0] = (_Float16)bval;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {5.96e-8f, 1.0f}, {1.0f, 3.0e-5f}, {3.0e-5f, 1024.0f}, {6.2e-5f, 1.0f}};
  for (auto& cs : cases) {
    probe<<<1, 64>>>(d, cs[0], cs[1]);
    float h = 0.f;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    const float want = 4.0f * (float)(_Float16)cs[0] * (float)(_Float16)cs[1];
    printf("a = %.3e (fp16 %.6e)  b = %.3e : mfma gives %.6e, exact %.6e  -> %s\n", cs[0], (double)(float)(_Float16)cs[0], cs[1], h, want,
           h == want ? "kept" : (h == 0.f ? "FLUSHED" : "other"));
  }
  return 0;
}
