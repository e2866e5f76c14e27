// Data-parallel exchange of sampled token ids: a thin C ABI over RCCL for hosts that do not go
// through torch.distributed.  One process per GPU; the caller distributes the 128-byte unique id
// of rank 0 to the other ranks by whatever channel it has (a file, a socket, a key-value store).
//
// RCCL is bound at run time (dlopen of librccl.so.1, which resolves to the copy already loaded
// in the process if there is one), so the library itself loads on machines without it and the
// entry points report MI355_EUNSUPPORTED there.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace mi355 {
namespace {

constexpr int kIdBytes = 128;                       // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kIdBytes]; };       // passed by value to ncclCommInitRank
constexpr int kNcclInt32 = 2;                       // ncclInt32

struct Rccl {
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
  });
  return r;
}

int rccl_fail(const char* what, int rc) {
  const Rccl& r = rccl();
  return fail(MI355_ECOLLECTIVE, "%s: %s (rccl result %d)", what, r.GetErrorString ? r.GetErrorString(rc) : "error", rc);
}

// A device-started exchange: the collective sits on a side stream of its own behind a one-wave
// "watcher" kernel that returns once every id of the step has been written.  The producer's
// stream carries nothing for the exchange: no event record, no wait, no extra launch, and the
// sampler is the plain one.  Ids are non-negative; a send buffer holds kNotWritten in every entry
// the watcher has not seen written yet, and is marked again right after its collective.
constexpr int kExchangeSlots = 8;
constexpr int32_t kNotWritten = -1;
constexpr long long kWatchTicks = 10ll * 100000000ll;   // the watcher gives up after 10 s (100 MHz clock)

struct Exchange {
  void* comm = nullptr;
  int device = -1;
  hipStream_t side = nullptr;
  int* gave_up = nullptr;                   // pinned host memory, raised by a watcher that timed out
  hipEvent_t done[kExchangeSlots] = {};
  hipEvent_t ready[kExchangeSlots] = {};    // recorded on the producer's stream (no system fence)
  bool recorded[kExchangeSlots] = {};
};

// One wave.  Agent-scope loads go past the XCD's L2 to device-coherent memory, where the
// producer's ids land when its kernel ends; between polls the wave sleeps.
__global__ void __launch_bounds__(64) watch_ids_kernel(const int32_t* ids, int64_t count, int* gave_up, int naps) {
  const long long t0 = wall_clock64();
  for (;;) {
    bool written = true;
    for (int64_t i = threadIdx.x; i < count; i += 64)
      written &= __hip_atomic_load(ids + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kNotWritten;
    if (__all(written)) return;
    if (wall_clock64() - t0 > kWatchTicks) {
      if (threadIdx.x == 0) __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (naps == 0) __builtin_amdgcn_s_sleep(32);
    for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(127);
  }
}

__global__ void mark_ids_kernel(int32_t* ids, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) ids[i] = kNotWritten;
}

int hip_fail(const char* what, hipError_t e) {
  fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

int mark(Exchange* x, int32_t* ids, int64_t count) {
  hipLaunchKernelGGL(mark_ids_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, x->side, ids, count);
  return check_launch("mark_ids_kernel");
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_dp_unique_id(void* id_out) {
  if (!id_out) return fail(MI355_EINVAL, "dp_unique_id: null pointer");
  const Rccl& r = rccl();
  if (!r.ok) return fail(MI355_EUNSUPPORTED, "dp_unique_id: librccl.so.1 is not available");
  if (int rc = r.GetUniqueId(static_cast<UniqueId*>(id_out))) return rccl_fail("ncclGetUniqueId", rc);
  return 0;
}

extern "C" int mi355_dp_init(const void* id, int32_t nranks, int32_t rank, void** comm_out) {
  if (!id || !comm_out) return fail(MI355_EINVAL, "dp_init: null pointer");
  if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(MI355_EINVAL, "dp_init: rank %d of %d", rank, nranks);
  const Rccl& r = rccl();
  if (!r.ok) return fail(MI355_EUNSUPPORTED, "dp_init: librccl.so.1 is not available");
  UniqueId uid;
  memcpy(uid.internal, id, kIdBytes);
  void* comm = nullptr;
  if (int rc = r.CommInitRank(&comm, nranks, uid, rank)) return rccl_fail("ncclCommInitRank", rc);
  *comm_out = comm;
  return 0;
}

extern "C" int mi355_dp_allgather_i32(void* comm, const int32_t* send, int32_t* recv, int64_t count, void* stream) {
  if (count == 0) return 0;
  if (!comm || !send || !recv || count < 0) return fail(MI355_EINVAL, "dp_allgather_i32: bad argument");
  const Rccl& r = rccl();
  if (!r.ok) return fail(MI355_EUNSUPPORTED, "dp_allgather_i32: librccl.so.1 is not available");
  if (int rc = r.AllGather(send, recv, (size_t)count, kNcclInt32, comm, static_cast<hipStream_t>(stream)))
    return rccl_fail("ncclAllGather", rc);
  return 0;
}

extern "C" int mi355_dp_destroy(void* comm) {
  if (!comm) return 0;
  const Rccl& r = rccl();
  if (!r.ok) return fail(MI355_EUNSUPPORTED, "dp_destroy: librccl.so.1 is not available");
  if (int rc = r.CommDestroy(comm)) return rccl_fail("ncclCommDestroy", rc);
  return 0;
}

extern "C" int mi355_dp_exchange_create(void* comm, void** exchange_out) {
  if (!comm || !exchange_out) return fail(MI355_EINVAL, "dp_exchange_create: null pointer");
  Exchange* x = new Exchange;
  x->comm = comm;
  hipError_t e = hipGetDevice(&x->device);
  if (!e) {
    // A stream created WITH A PRIORITY (lowest: its kernels wait for ids and move a few hundred
    // bytes) gets a hardware queue of its own class.  On a plain second stream a resident watcher --
    // or a blocked barrier packet -- slowed the compute stream's workgroup dispatch: a decode launch
    // of 2048 workgroups ran 3 us longer (163 -> 166 us), one of 256 long workgroups did not; with
    // the priority stream the step costs 0.3-0.5 us more than with no exchange at all, at either end
    // of the priority range.
    int least = 0, greatest = 0;
    e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (!e) e = hipStreamCreateWithPriority(&x->side, hipStreamNonBlocking, least);
  }
  if (!e) e = hipHostMalloc(reinterpret_cast<void**>(&x->gave_up), sizeof(int), hipHostMallocMapped);
  if (!e) *x->gave_up = 0;
  for (int i = 0; i < kExchangeSlots && !e; ++i) e = hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming);
  for (int i = 0; i < kExchangeSlots && !e; ++i)
    e = hipEventCreateWithFlags(&x->ready[i], hipEventDisableTiming | hipEventDisableSystemFence);
  if (e) {
    mi355_dp_exchange_destroy(x);
    return hip_fail("dp_exchange_create", e);
  }
  *exchange_out = x;
  return 0;
}

extern "C" int mi355_dp_exchange_arm(void* exchange, int32_t* send, int64_t count) {
  Exchange* x = static_cast<Exchange*>(exchange);
  if (!x || !send || count <= 0) return fail(MI355_EINVAL, "dp_exchange_arm: bad argument");
  if (int rc = mark(x, send, count)) return rc;
  if (hipError_t e = hipStreamSynchronize(x->side)) return hip_fail("hipStreamSynchronize", e);
  return 0;
}

extern "C" int mi355_dp_exchange_launch(void* exchange, int32_t* send, int32_t* recv, int64_t count, int32_t slot) {
  Exchange* x = static_cast<Exchange*>(exchange);
  if (!x || !send || !recv || count <= 0 || slot < 0 || slot >= kExchangeSlots)
    return fail(MI355_EINVAL, "dp_exchange_launch: bad argument");
  const Rccl& r = rccl();
  if (!r.ok) return fail(MI355_EUNSUPPORTED, "dp_exchange_launch: librccl.so.1 is not available");
  const int probe = knob(kKnobDpExchangeProbe);
  if (!(probe & 4)) {
    hipLaunchKernelGGL(watch_ids_kernel, dim3(1), dim3(64), 0, x->side, send, count, x->gave_up, (probe >> 8) & 0xff);
    if (int rc = check_launch("watch_ids_kernel")) return rc;
  }
  if (!(probe & 1))
    if (int rc = r.AllGather(send, recv, (size_t)count, kNcclInt32, x->comm, x->side)) return rccl_fail("ncclAllGather", rc);
  if (int rc = mark(x, send, count)) return rc;
  if (probe & 2) return 0;
  if (hipError_t e = hipEventRecord(x->done[slot], x->side)) return hip_fail("hipEventRecord", e);
  x->recorded[slot] = true;
  return 0;
}

extern "C" int mi355_dp_exchange_launch_after(void* exchange, void* producer_stream, const int32_t* send,
                                              int32_t* recv, int64_t count, int32_t slot) {
  Exchange* x = static_cast<Exchange*>(exchange);
  if (!x || !send || !recv || count <= 0 || slot < 0 || slot >= kExchangeSlots)
    return fail(MI355_EINVAL, "dp_exchange_launch_after: bad argument");
  const Rccl& r = rccl();
  if (!r.ok) return fail(MI355_EUNSUPPORTED, "dp_exchange_launch_after: librccl.so.1 is not available");
  const int probe = knob(kKnobDpExchangeProbe);
  if (!(probe & 8))
    if (hipError_t e = hipEventRecord(x->ready[slot], static_cast<hipStream_t>(producer_stream))) return hip_fail("hipEventRecord", e);
  if (!(probe & 12))
    if (hipError_t e = hipStreamWaitEvent(x->side, x->ready[slot], 0)) return hip_fail("hipStreamWaitEvent", e);
  if (!(probe & 1))
    if (int rc = r.AllGather(send, recv, (size_t)count, kNcclInt32, x->comm, x->side)) return rccl_fail("ncclAllGather", rc);
  if (probe & 2) return 0;
  if (hipError_t e = hipEventRecord(x->done[slot], x->side)) return hip_fail("hipEventRecord", e);
  x->recorded[slot] = true;
  return 0;
}

static int exchange_gave_up(Exchange* x) {
  if (!*reinterpret_cast<volatile int*>(x->gave_up)) return 0;
  return fail(MI355_ECOLLECTIVE, "dp_exchange: a step's ids were not written within 10 s of its exchange being due; "
                                 "what was gathered for it is not valid");
}

extern "C" int mi355_dp_exchange_done(void* exchange, int32_t slot) {
  Exchange* x = static_cast<Exchange*>(exchange);
  if (!x || slot < 0 || slot >= kExchangeSlots) return fail(MI355_EINVAL, "dp_exchange_done: bad argument");
  if (!x->recorded[slot]) return 1;
  const hipError_t e = hipEventQuery(x->done[slot]);
  if (e == hipErrorNotReady) return 0;
  if (e != hipSuccess) return fail(MI355_ECOLLECTIVE, "dp_exchange_done: %s", hipGetErrorString(e));
  if (int rc = exchange_gave_up(x)) return rc;
  return 1;
}

extern "C" int mi355_dp_exchange_wait(void* exchange, int32_t slot, void* stream, int32_t host) {
  Exchange* x = static_cast<Exchange*>(exchange);
  if (!x || slot < 0 || slot >= kExchangeSlots) return fail(MI355_EINVAL, "dp_exchange_wait: bad argument");
  if (!x->recorded[slot]) return 0;
  if (host) {
    if (hipError_t e = hipEventSynchronize(x->done[slot])) return hip_fail("hipEventSynchronize", e);
    return exchange_gave_up(x);
  }
  if (hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), x->done[slot], 0))
    return hip_fail("hipStreamWaitEvent", e);
  return 0;
}

extern "C" int mi355_dp_exchange_destroy(void* exchange) {
  Exchange* x = static_cast<Exchange*>(exchange);
  if (!x) return 0;
  if (x->side) (void)hipStreamSynchronize(x->side);    // a watcher still waiting returns within its time limit
  for (int i = 0; i < kExchangeSlots; ++i)
    if (x->done[i]) (void)hipEventDestroy(x->done[i]);
  for (int i = 0; i < kExchangeSlots; ++i)
    if (x->ready[i]) (void)hipEventDestroy(x->ready[i]);
  if (x->gave_up) (void)hipHostFree(x->gave_up);
  if (x->side) (void)hipStreamDestroy(x->side);
  delete x;
  return 0;
}
