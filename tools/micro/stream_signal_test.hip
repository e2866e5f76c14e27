// What it costs a compute stream to tell a side stream "this kernel's output is ready":
//   (1) hipEventRecord + hipStreamWaitEvent (default event),
//   (2) the same with hipEventDisableSystemFence,
//   (3) nothing on the compute stream: the kernel itself raises a word of signal memory and the side
//       stream waits for the value (hipStreamWaitValue64, served by the command processor).
// Also checks that (3) orders correctly: the side stream's kernel must see the producer's data.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/stream_signal_test.hip -o /tmp/stream_signal_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void producer(long long ticks, int* data, int value, unsigned long long* flag, unsigned long long raise_to) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __hip_atomic_store(data, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flag) {
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_store(flag, raise_to, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void consumer(const int* data, int want, int* errors, int* seen) {
  int v = __hip_atomic_load(data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v < want) atomicAdd(errors, 1);
  *seen = v;
}

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  int *data, *errors, *seen;
  CK(hipMalloc(&data, 4)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&seen, 4));
  CK(hipMemset(data, 0, 4)); CK(hipMemset(errors, 0, 4));
  unsigned long long* flag = nullptr;
  if (can) {
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    *(volatile unsigned long long*)flag = 0;     // signal memory is host-visible
  }
  const int iters = 300;
  const long long ticks = 5000;   // 50 us at 100 MHz
  hipEvent_t ev_default, ev_light;
  CK(hipEventCreateWithFlags(&ev_default, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_light, hipEventDisableTiming | hipEventDisableSystemFence));
  for (int mode = 0; mode <= (can ? 3 : 2); ++mode) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= iters; ++i) {
      int value = mode * 1000 + i;
      unsigned long long level = (unsigned long long)(mode == 3 ? i : 0);
      hipLaunchKernelGGL(producer, dim3(256), dim3(64), 0, a, ticks, data, value, mode == 3 ? flag : nullptr, level);
      if (mode == 1 || mode == 2) {
        hipEvent_t ev = mode == 1 ? ev_default : ev_light;
        CK(hipEventRecord(ev, a));
        CK(hipStreamWaitEvent(b, ev, 0));
      } else if (mode == 3) {
        CK(hipStreamWaitValue64(b, flag, level, hipStreamWaitValueGte, ~0ull));
      }
      if (mode) hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, data, value, errors, seen);
    }
    CK(hipStreamSynchronize(a));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(b));
    int herr = 0, hseen = 0;
    CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hseen, seen, 4, hipMemcpyDeviceToHost));
    CK(hipMemset(errors, 0, 4));
    const char* names[] = {"no signalling", "event record + wait (default event)", "event record + wait (no system fence)",
                           "kernel raises signal memory, side stream waits for the value"};
    printf("%-62s %8.2f us per iteration on the compute stream; consumer errors %d (last seen %d)\n", names[mode],
           std::chrono::duration<double, std::micro>(t1 - t0).count() / iters, herr, hseen);
  }
  return 0;
}
