// Does a SIMD overlap MFMA execution with VALU work?  Four kernels, same launch geometry
// (one wave per SIMD unless noted): MFMA only, VALU only, both interleaved in one wave, and
// two waves per SIMD with one doing MFMA and the other VALU.  Prints cycles per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kIters = 2000, kMfma = 8, kValu = 64;

template <int MODE>  // 0 mfma, 1 valu, 2 both in one wave, 3 split by wave parity
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, float seed) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  f32x16 acc[kMfma];
  for (int m = 0; m < kMfma; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) + threadIdx.x;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 0);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 1);
  const long long t0 = clock64();
  for (int it = 0; it < kIters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int m = 0; m < kMfma; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int j = 0; j < kValu; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int m = 0; m < kMfma; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  float* out; long long* cyc; long long h = 0;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float)); (void)hipMalloc(&cyc, 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads/WG %4d: %8.3f ms, %7.1f clock64 ticks per iteration (%d MFMA 32x32x16 + %d v_fma per wave)\n", name, threads, ms,
         (double)h / kIters, kMfma, kValu);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("MFMA only", 256);
  run<1>("VALU only", 256);
  run<2>("MFMA + VALU in the same wave", 256);
  run<0>("MFMA only, 2 waves per SIMD", 512);
  run<1>("VALU only, 2 waves per SIMD", 512);
  run<3>("waves 0-3 MFMA, waves 4-7 VALU (1+1 per SIMD)", 512);
  run<2>("MFMA + VALU in every wave, 2 per SIMD", 512);
  return 0;
}
