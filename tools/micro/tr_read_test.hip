// Micro check of ds_read_b64_tr_b16 lane semantics on gfx950 (exact integer data).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];  // [row][col], 64 cols
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;  // value = row*64 + col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // group g reads block rows 4g..4g+3, cols 0..15 : lane 4q+p gives address of row q, cols 4p..4p+3
  const short* addr = lds + (4 * g + q) * 64 + 4 * p;
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = t[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
    int g = lane >> 4, i = lane & 15;
    int want = (4 * g + e) * 64 + i;   // lane i receives column i of the 4 rows, row e in element e
    if (h[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", lane, e, h[lane*4+e], want); ++bad; }
  }
  printf("tr_read semantics: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as documented", bad);
  return 0;
}
