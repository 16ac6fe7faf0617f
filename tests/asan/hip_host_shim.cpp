// TEST INFRASTRUCTURE -- never part of the product library.
//
// Host-memory stand-ins for the ~25 HIP runtime entry points libafx's HOST side calls, so that the C++ around the
// kernels (weight store, name parsing, workspace carving, launch-argument arithmetic, the ragged-batch bucket copies,
// taps, profiler) can run under AddressSanitizer / UndefinedBehaviorSanitizer on a machine WITHOUT a GPU (GPU ASan is not
// available on the pool).  `make -C real-time-deepfake-speech-detection_amd/csrc asan` compiles the
// same sources with `--cuda-host-only -fsanitize=address,undefined`, links them with THIS file into lib/libafx_asan.so
// (-Bsymbolic: the library's HIP calls bind here) and tests/test_cpu_asan.py drives it through the C ABI.
//   device memory  = malloc'd host memory (so every hipMemcpy / hipMemset the host side issues is bounds-checked by ASan)
//   kernel launch  = nothing (outputs are garbage; what is exercised is the code that decides WHAT to launch and WHERE)
// libafx.so never links this file, and nothing here computes a result: it is not a CPU path of the product.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipRuntimeGetVersion(int* v) { *v = HIP_VERSION; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_host_shim: error"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

struct ShimEvent { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new ShimEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete (ShimEvent*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { ((ShimEvent*)e)->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(((ShimEvent*)b)->t - ((ShimEvent*)a)->t).count();
  return hipSuccess;
}

// kernel launches: the <<<>>> expansion of the host pass
static thread_local dim3 g_grid, g_block;
static thread_local size_t g_shmem;
static thread_local hipStream_t g_stream;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s) {
  g_grid = grid; g_block = block; g_shmem = shmem; g_stream = s;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* s) {
  *grid = g_grid; *block = g_block; *shmem = g_shmem; *s = g_stream;
  return hipSuccess;
}
long g_shim_launches = 0;  // (read by the test: the forward did reach its launches)
hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t, hipStream_t) {
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || block.x * block.y * block.z > 1024) return hipErrorInvalidConfiguration;
  ++g_shim_launches;
  return hipSuccess;
}
long afx_shim_launch_count(void) { return g_shim_launches; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

}  // extern "C"
