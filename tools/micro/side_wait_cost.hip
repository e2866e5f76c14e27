// What a side stream that waits for "step done" costs a compute stream that runs three kernels per
// step (a short one, a chip-filling one in eight rounds of workgroups, a 9 us one), by the way
// the side stream waits:
//   0 nothing on the side stream
//   1 hipStreamWaitValue64 on signal memory (host memory, polled by the command processor)
//   2 the same followed by a one-thread kernel
//   3 a one-wave kernel that polls a flag in DEVICE memory (sleeping between polls, bounded)
//   4 event record (no system fence) on the compute stream + hipStreamWaitEvent + one-thread kernel
//   hipcc --offload-arch=gfx950 -O2 tools/micro/side_wait_cost.hip -o tools/_ab/side_wait_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks, unsigned long long* flag_host, unsigned long long* flag_dev, unsigned long long level,
                     unsigned int* arrivals) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (arrivals && threadIdx.x == 0) {
    unsigned int before = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1 == gridDim.x) {
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (flag_dev) __hip_atomic_store(flag_dev, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (flag_host) __hip_atomic_store(flag_host, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void tiny(int* p) { p[0] += 1; }

__global__ void poll(const unsigned long long* flag_dev, unsigned long long level, int* timed_out) {
  long long t0 = wall_clock64();
  while (__hip_atomic_load(flag_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < level) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 200000000ll) { *timed_out = 1; break; }   // 2 s
  }
}

int main() {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  int* scratch; unsigned int* arrivals; unsigned long long *flag_dev, *flag_host; int* timed_out;
  CK(hipMalloc(&scratch, 64)); CK(hipMemset(scratch, 0, 64));
  CK(hipMalloc(&arrivals, 4)); CK(hipMemset(arrivals, 0, 4));
  CK(hipMalloc(&flag_dev, 8)); CK(hipMemset(flag_dev, 0, 8));
  CK(hipMalloc(&timed_out, 4)); CK(hipMemset(timed_out, 0, 4));
  CK(hipExtMallocWithFlags((void**)&flag_host, 8, hipMallocSignalMemory));
  *(volatile unsigned long long*)flag_host = 0;
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  const int iters = 300;
  const char* names[] = {"nothing on the side stream", "wait for a value in signal memory", "wait for a value + one-thread kernel",
                         "one-wave kernel polling device memory", "event record (no fence) + stream wait + one-thread kernel"};
  unsigned long long level = 0;
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 5; ++mode) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= iters; ++i) {
      ++level;
      hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, 300ll, nullptr, nullptr, 0ull, nullptr);          // 3 us
      hipLaunchKernelGGL(spin, dim3(2048), dim3(512), 0, a, 2000ll, nullptr, nullptr, 0ull, nullptr);      // 8 rounds x 20 us
      hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, a, 800ll, (mode == 1 || mode == 2) ? flag_host : nullptr,
                         mode == 3 ? flag_dev : nullptr, level, mode ? arrivals : nullptr);               // 8 us
      if (mode == 1 || mode == 2) CK(hipStreamWaitValue64(b, flag_host, level, hipStreamWaitValueGte, ~0ull));
      if (mode == 3) hipLaunchKernelGGL(poll, dim3(1), dim3(64), 0, b, flag_dev, level, timed_out);
      if (mode == 4) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); }
      if (mode == 2 || mode == 4) hipLaunchKernelGGL(tiny, dim3(1), dim3(1), 0, b, scratch);
    }
    CK(hipStreamSynchronize(a));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(b));
    int to = 0;
    CK(hipMemcpy(&to, timed_out, 4, hipMemcpyDeviceToHost));
    printf("%-60s %8.2f us per step%s\n", names[mode], std::chrono::duration<double, std::micro>(t1 - t0).count() / iters,
           to ? "  (a poll timed out)" : "");
  }
  return 0;
}
