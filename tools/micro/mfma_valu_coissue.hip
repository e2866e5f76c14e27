// One wave per SIMD: how many independent VALU instructions fit between the matrix instructions
// of a wave before they cost time?  Per variant: a loop of 32x32x16 bf16 matrix instructions
// (a dependent chain on one accumulator, or four accumulators in rotation) with K independent
// v_fma_f32 (or v_exp_f32) after each; prints shader-clock ticks per matrix instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu_coissue tools/micro/mfma_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int CHAINS, bool EXP>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* ticks, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x16(0.f);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % CHAINS]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 8]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k % 8]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int K, int CHAINS, bool EXP>
void run(float* out, unsigned long long* ticks) {
  const int iters = 2000;
  probe<K, CHAINS, EXP><<<256, 256>>>(out, ticks, iters);
  probe<K, CHAINS, EXP><<<256, 256>>>(out, ticks, iters);
  hipDeviceSynchronize();
  unsigned long long t;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("chains %d  %s x %d per matrix instruction: %.1f ticks per matrix instruction\n", CHAINS, EXP ? "v_exp" : "v_fma", K,
         (double)t / (iters * 8.0));
}

int main() {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&ticks, 8);
  run<0, 1, false>(out, ticks); run<2, 1, false>(out, ticks); run<4, 1, false>(out, ticks); run<6, 1, false>(out, ticks); run<8, 1, false>(out, ticks);
  run<0, 4, false>(out, ticks); run<2, 4, false>(out, ticks); run<4, 4, false>(out, ticks); run<6, 4, false>(out, ticks); run<8, 4, false>(out, ticks);
  run<1, 1, true>(out, ticks); run<2, 1, true>(out, ticks); run<4, 1, true>(out, ticks);
  return 0;
}
