// stonk_comm_*: the gradient exchange of the data-parallel step behind the C ABI: RCCL
// collectives on a stream the LIBRARY owns, handed over by events - the caller's compute stream never waits on the host
// and never runs a collective itself.
//
//   producer stream --(event)--> comm stream: ncclAllReduce / ReduceScatter / AllGather --(event)--> consumer stream
//
// What this replaces: torch DistributedDataParallel's bucketed all-reduce, which the reference gets from HF `Trainer`
// when launched distributed (ref:src/stonkgs/models/stonkgs_pretraining.py:215-223), and DeepSpeed ZeRO-2's
// reduce-scatter / all-gather when `deepspeed=True` (:174-175). A maintainer who binds the C ABI gets
// the exchange from here; the Python package can run on it (`TrainingArguments.comm_backend = "stonk"`) or on
// torch.distributed (the default, whose process group the rest of a PyTorch program already has).
//
// RCCL is resolved at run time (dlopen / dlsym of the five entry points used): the kernel library stays loadable - and
// every other entry point usable - on a machine without RCCL, and a process that already carries RCCL (PyTorch's copy)
// reuses that one instead of loading a second.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>

#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {   // a copy the process already holds first (RTLD_NOLOAD), then the system's
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (r.handle) break;
    }
    for (int i = 0; !r.handle && i < 2; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(r.handle, "ncclReduceScatter");
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather;
  });
  return r;
}

struct Comm {
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;   // the collectives' own stream
  hipEvent_t handoff = nullptr;   // producer -> comm
  hipEvent_t done = nullptr;      // comm -> consumer
  int world = 0, rank = 0, device = 0;
};

inline int hip_status(hipError_t e) { return e == hipSuccess ? STONK_OK : (int)e; }
// RCCL failures are reported in the hipError range's upper end so that they cannot be taken for a kernel's error code
inline int nccl_status(ncclResult_t r) { return r == ncclSuccess ? STONK_OK : 10000 + (int)r; }

inline bool dtype_of(int dtype, ncclDataType_t& t) {
  if (dtype == 0) t = ncclFloat32;
  else if (dtype == 1) t = ncclBfloat16;
  else return false;
  return true;
}

// order the comm stream behind what `after_stream` has enqueued so far
int follow(Comm* c, void* after_stream) {
  hipError_t e = hipEventRecord(c->handoff, (hipStream_t)after_stream);
  if (e != hipSuccess) return (int)e;
  return hip_status(hipStreamWaitEvent(c->stream, c->handoff, 0));
}

}  // namespace

extern "C" int stonk_comm_unique_id(void* id_out) {
  STONK_CHECK_ARG(id_out, STONK_EINVAL);
  Rccl& r = rccl();
  if (!r.ok) return STONK_EINVAL;
  return nccl_status(r.GetUniqueId((ncclUniqueId*)id_out));
}

extern "C" int stonk_comm_init(void** comm_out, int world, int rank, const void* unique_id, int device) {
  STONK_CHECK_ARG(comm_out && unique_id && world >= 1 && rank >= 0 && rank < world && device >= 0, STONK_EINVAL);
  Rccl& r = rccl();
  if (!r.ok) return STONK_EINVAL;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  Comm* c = new (std::nothrow) Comm;
  if (!c) return STONK_EINVAL;
  c->world = world;
  c->rank = rank;
  c->device = device;
  ncclUniqueId id;
  __builtin_memcpy(&id, unique_id, sizeof(id));
  int rc = nccl_status(r.CommInitRank(&c->nccl, world, id, rank));
  if (rc == STONK_OK) rc = hip_status(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  if (rc == STONK_OK) rc = hip_status(hipEventCreateWithFlags(&c->handoff, hipEventDisableTiming));
  if (rc == STONK_OK) rc = hip_status(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  if (rc != STONK_OK) {
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->handoff) (void)hipEventDestroy(c->handoff);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->nccl) (void)r.CommDestroy(c->nccl);
    delete c;
    return rc;
  }
  *comm_out = c;
  return STONK_OK;
}

extern "C" int stonk_comm_allreduce_async(void* comm, void* buf, int64_t n, int dtype, void* after_stream) {
  Comm* c = (Comm*)comm;
  ncclDataType_t t;
  STONK_CHECK_ARG(c && buf && n >= 0 && dtype_of(dtype, t), STONK_EINVAL);
  if (n == 0) return STONK_OK;
  const int rc = follow(c, after_stream);
  if (rc != STONK_OK) return rc;
  return nccl_status(rccl().AllReduce(buf, buf, (size_t)n, t, ncclSum, c->nccl, c->stream));
}

extern "C" int stonk_comm_reduce_scatter_async(void* comm, const void* send, void* recv, int64_t recv_n, int dtype,
                                               void* after_stream) {
  Comm* c = (Comm*)comm;
  ncclDataType_t t;
  STONK_CHECK_ARG(c && send && recv && recv_n >= 0 && dtype_of(dtype, t), STONK_EINVAL);
  if (recv_n == 0) return STONK_OK;
  const int rc = follow(c, after_stream);
  if (rc != STONK_OK) return rc;
  return nccl_status(rccl().ReduceScatter(send, recv, (size_t)recv_n, t, ncclSum, c->nccl, c->stream));
}

extern "C" int stonk_comm_allgather_async(void* comm, const void* send, void* recv, int64_t send_n, int dtype,
                                          void* after_stream) {
  Comm* c = (Comm*)comm;
  ncclDataType_t t;
  STONK_CHECK_ARG(c && send && recv && send_n >= 0 && dtype_of(dtype, t), STONK_EINVAL);
  if (send_n == 0) return STONK_OK;
  const int rc = follow(c, after_stream);
  if (rc != STONK_OK) return rc;
  return nccl_status(rccl().AllGather(send, recv, (size_t)send_n, t, c->nccl, c->stream));
}

extern "C" int stonk_comm_wait(void* comm, void* stream) {
  Comm* c = (Comm*)comm;
  STONK_CHECK_ARG(c, STONK_EINVAL);
  hipError_t e = hipEventRecord(c->done, c->stream);
  if (e != hipSuccess) return (int)e;
  return hip_status(hipStreamWaitEvent((hipStream_t)stream, c->done, 0));
}

extern "C" void* stonk_comm_stream(void* comm) { return comm ? (void*)((Comm*)comm)->stream : nullptr; }

extern "C" int stonk_comm_destroy(void* comm) {
  Comm* c = (Comm*)comm;
  STONK_CHECK_ARG(c, STONK_EINVAL);
  (void)hipStreamSynchronize(c->stream);
  const int rc = nccl_status(rccl().CommDestroy(c->nccl));
  (void)hipEventDestroy(c->done);
  (void)hipEventDestroy(c->handoff);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return rc;
}
