// comm.hip — the thin RCCL layer of librcdm_hip.so: a communicator handle plus broadcast and all-gather of raw device
// bytes on a caller-given HIP stream (graph-capturable like every other entry point).
//
// Reference: the reference has NO inter-GPU communication — stage2_batchtest_rcdms_model.py:457-468 spawns one process
// per device and every process loads every checkpoint itself.  What these entry points are for (SURVEY section 8(b),(e)):
//   * rcdm_bcast      — rank 0 packs the weights once, the other ranks receive the packed f16 images over xGMI;
//   * rcdm_allgather  — the CFG-split latency mode: two GPUs per story, each evaluates ONE classifier-free-guidance
//     half of the UNet (batch elements are independent inside it), then the two noise predictions (164 KB each at
//     64x64 latents) are exchanged inside the step graph and both ranks apply the same CFG + DDIM update
//     (RCDMs_pipeline.py:482-497 with the two halves of `latent_model_input` on two devices).
// librccl is opened lazily with dlopen on the first call: a single-GPU process never maps it.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

// the few RCCL declarations this file needs (ABI of rccl.h, ROCm 7: ncclUniqueId = 128 opaque bytes passed by value,
// ncclComm_t = opaque pointer, ncclResult_t 0 = success, ncclDataType_t ncclUint8 = 1)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
constexpr int kUint8 = 1;

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  BroadcastFn broadcast = nullptr;
  AllGatherFn all_gather = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
    x.get_unique_id = (GetUniqueIdFn)dlsym(x.lib, "ncclGetUniqueId");
    x.comm_init_rank = (CommInitRankFn)dlsym(x.lib, "ncclCommInitRank");
    x.comm_destroy = (CommDestroyFn)dlsym(x.lib, "ncclCommDestroy");
    x.broadcast = (BroadcastFn)dlsym(x.lib, "ncclBroadcast");
    x.all_gather = (AllGatherFn)dlsym(x.lib, "ncclAllGather");
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.broadcast && x.all_gather;
    return x;
  }();
  return r;
}

// what rcdm_comm_create hands out: the RCCL communicator, its size (root checks) and the device it was bound to
struct Handle { Comm comm; int nranks; int device; };
inline bool on_own_device(const Handle* h) {
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess && dev == h->device;
}

thread_local int g_last_rccl_result = 0;
inline int rc(int nccl_result) {
  if (nccl_result == 0) return RCDM_OK;
  g_last_rccl_result = nccl_result;
  return RCDM_ECOMM;
}

}  // namespace

extern "C" {

int rcdm_comm_last_error(void) { return g_last_rccl_result; }

int rcdm_comm_unique_id(void* id128) {
  if (!id128) return RCDM_EINVAL;
  if (!rccl().ok) return RCDM_ECOMM;
  UniqueId id;
  memset(&id, 0, sizeof id);
  const int r = rc(rccl().get_unique_id(&id));
  if (r == RCDM_OK) memcpy(id128, &id, sizeof id);
  return r;
}

int rcdm_comm_create(const void* id128, int32_t nranks, int32_t rank, void** comm) {
  if (!id128 || !comm || nranks <= 0 || rank < 0 || rank >= nranks) return RCDM_EINVAL;
  *comm = nullptr;
  if (!rccl().ok) return RCDM_ECOMM;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return RCDM_ELAUNCH;   // RCCL binds the communicator to the CURRENT device
  UniqueId id;
  memcpy(&id, id128, sizeof id);
  Comm c = nullptr;
  const int r = rc(rccl().comm_init_rank(&c, nranks, id, rank));
  if (r == RCDM_OK) *comm = new Handle{c, nranks, dev};
  return r;
}

int rcdm_comm_destroy(void* comm) {
  if (!comm) return RCDM_EINVAL;
  if (!rccl().ok) return RCDM_ECOMM;
  Handle* h = (Handle*)comm;
  const int r = rc(rccl().comm_destroy(h->comm));
  delete h;
  return r;
}

int rcdm_bcast(void* comm, void* buf, size_t bytes, int32_t root, void* stream) {
  Handle* h = (Handle*)comm;
  if (!h || !buf || root < 0 || root >= h->nranks || !on_own_device(h)) return RCDM_EINVAL;
  if (bytes == 0) return RCDM_OK;
  if (!rccl().ok) return RCDM_ECOMM;
  return rc(rccl().broadcast(buf, buf, bytes, kUint8, root, h->comm, (hipStream_t)stream));
}

int rcdm_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  Handle* h = (Handle*)comm;
  if (!h || !send || !recv || !on_own_device(h)) return RCDM_EINVAL;
  if (bytes_per_rank == 0) return RCDM_OK;
  if (!rccl().ok) return RCDM_ECOMM;
  return rc(rccl().all_gather(send, recv, bytes_per_rank, kUint8, h->comm, (hipStream_t)stream));
}

}  // extern "C"
