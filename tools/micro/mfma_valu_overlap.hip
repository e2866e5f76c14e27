// Does a SIMD overlap MFMA execution with VALU work?  Same launch geometry throughout (one wave
// per SIMD unless noted): MFMA only, VALU only, both in one wave as the compiler orders them
// (all MFMAs, then all VALU), both in one wave with the VALU instructions placed between the
// MFMAs (sched_group_barrier: 1 MFMA, then kValu/kMfma VALU), and two waves per SIMD with one
// doing MFMA and the other VALU.  Prints cycles per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kIters = 2000, kMfma = 8;

template <int MODE, int kValu>  // 0 mfma, 1 valu, 2 both in one wave, 3 split by wave parity, 4 = 2 with placement
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, float seed) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  f32x16 acc[kMfma];
  for (int m = 0; m < kMfma; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) + threadIdx.x;
  const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && ((wave >> 2) & 1) == 0);
  const bool do_valu = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && ((wave >> 2) & 1) == 1);
  const float lo = seed * 0.25f, hi = seed * 1.0e6f;
  const long long t0 = clock64();
  for (int it = 0; it < kIters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int m = 0; m < kMfma; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int j = 0; j < kValu; ++j) v[j & 7] = __builtin_amdgcn_fmed3f(v[j & 7], lo, hi);   // v_med3_f32: no packed form for the compiler to fuse pairs into
    }
    if (MODE == 4) {
#pragma unroll
      for (int m = 0; m < kMfma; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, kValu / kMfma, 0);
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int m = 0; m < kMfma; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int kValu>
void run(const char* name, int threads) {
  float* out; long long* cyc; long long h = 0;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float)); (void)hipMalloc(&cyc, 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, kValu>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, kValu>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads/WG %4d: %8.3f ms, %7.1f clock64 ticks per iteration (%d MFMA 32x32x16 + %d v_med3 per wave)\n", name, threads, ms,
         (double)h / kIters, kMfma, kValu);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 64>("MFMA only", 256);
  run<1, 64>("VALU only", 256);
  run<2, 64>("MFMA then VALU in the same wave", 256);
  run<4, 64>("same, 8 v_med3 placed after each MFMA", 256);
  run<1, 40>("VALU only", 256);
  run<2, 40>("MFMA then VALU in the same wave", 256);
  run<4, 40>("same, 5 v_med3 placed after each MFMA", 256);
  run<1, 24>("VALU only", 256);
  run<4, 24>("3 v_med3 placed after each MFMA", 256);
  run<0, 64>("MFMA only, 2 waves per SIMD", 512);
  run<1, 64>("VALU only, 2 waves per SIMD", 512);
  run<3, 64>("waves 0-3 MFMA, waves 4-7 VALU (1+1 per SIMD)", 512);
  run<2, 64>("MFMA then VALU in every wave, 2 per SIMD", 512);
  run<4, 40>("5 v_med3 after each MFMA, 2 per SIMD", 512);
  return 0;
}
