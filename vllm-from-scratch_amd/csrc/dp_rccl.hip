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
