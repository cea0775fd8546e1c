// comm.hip -- the data-parallel exchange step of the hot path, directly on RCCL (xGMI on an MI355X node).
//
// Replaces (reference, /root/reference): nn.parallel.DistributedDataParallel(model, device_ids, output_device)
// at trainer.py:79-82 (bucketed gradient all-reduce overlapped with backward, parameter broadcast at
// construction, buffer broadcast) and the nn.SyncBatchNorm reductions of main.py:190-191.
//
// One communicator handle per process (one process per GPU).  The handle owns a dedicated high-priority
// HIP stream for the bucket all-reduces and a ring of timing-less events:
//   cn_comm_allreduce_bucket(buf, n, after_a, after_b, n_after)
//       the communication stream waits for what is queued on the (up to two) producer streams NOW - the
//       bucket's last weight-gradient kernel on the wgrad side stream and its last BN/bias gradient on the
//       main stream - then runs ncclAllReduce(sum, fp32) in place.  Neither producer stream is stalled, so the
//       rest of backward overlaps the transfer.
//   cn_comm_join(stream)      `stream` waits for every bucket queued so far (before the optimizer step)
//   cn_comm_allreduce(stream) in-stream reduction (SyncBatchNorm statistics: on the critical path anyway)
// RCCL is bound at run time (dlopen of the librccl.so.1 the process already carries - PyTorch-ROCm ships
// one - else the ROCm installation's), so the kernel library itself has no link-time dependency on it.
// The library keeps no global state besides what a handle owns; the caller exchanges the 128-byte unique
// id between ranks (torch.distributed's store / any side channel) - rendezvous is not this library's job.
#include "cn_api_internal.h"
#include <stdio.h>
#include <stdlib.h>



#ifdef CN_EMULATE
// The TEST-ONLY CPU emulator has no devices and no RCCL: the host tests run the data-parallel path over gloo.
extern "C" int cn_comm_load(void) { cn_set_error("cn_comm: not available in the emulator build"); return CN_ERCCL; }
extern "C" int cn_comm_unique_id(char*) { cn_set_error("cn_comm: not available in the emulator build"); return CN_ERCCL; }
extern "C" int cn_comm_init(void**, const char*, int, int) { cn_set_error("cn_comm: not available in the emulator build"); return CN_ERCCL; }
extern "C" int cn_comm_info(void*, int*, int*, int*) { return CN_ERCCL; }
extern "C" int cn_comm_allreduce_bucket(void*, float*, long long, void*, void*, int) { return CN_ERCCL; }
extern "C" int cn_comm_join(void*, void*) { return CN_ERCCL; }
extern "C" int cn_comm_allreduce(void*, void*, long long, int, void*) { return CN_ERCCL; }
extern "C" int cn_comm_broadcast(void*, void*, long long, int, void*) { return CN_ERCCL; }
extern "C" int cn_comm_destroy(void*) { return CN_OK; }
int cn_comm_allreduce_bucket_issue(void*, float*, long long, void*, void*, int) { return CN_ERCCL; }
int cn_comm_join_issue(void*, void*) { return CN_ERCCL; }
int cn_comm_allreduce_issue(void*, void*, long long, int, void*) { return CN_ERCCL; }
#else
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

RcclApi g_api;   // function table only (no communicator state)

const RcclApi* rccl() {
  if (g_api.so != nullptr) return &g_api;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  const char* forced = getenv("CN_RCCL_LIB");   // explicit path: that file or nothing (also how the tests provoke a failed set-up)
  if (forced != nullptr && forced[0] != 0) {
    so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (so == nullptr) { cn_set_error("cn_comm: cannot load CN_RCCL_LIB=%s: %s", forced, dlerror()); return nullptr; }
  } else {
    so = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy the process already loaded (torch's)
    for (int i = 0; so == nullptr && i < 3; ++i) so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  }
  if (so == nullptr) { cn_set_error("cn_comm: cannot load librccl.so.1: %s", dlerror()); return nullptr; }
  RcclApi a;
  a.so = so;
#define CN_SYM(field, name)                                                   \
  *(void**)(&a.field) = dlsym(so, name);                                      \
  if (a.field == nullptr) { cn_set_error("cn_comm: %s missing in librccl", name); return nullptr; }
  CN_SYM(GetUniqueId, "ncclGetUniqueId")
  CN_SYM(CommInitRank, "ncclCommInitRank")
  CN_SYM(CommDestroy, "ncclCommDestroy")
  CN_SYM(AllReduce, "ncclAllReduce")
  CN_SYM(Broadcast, "ncclBroadcast")
  CN_SYM(GetErrorString, "ncclGetErrorString")
  CN_SYM(GetVersion, "ncclGetVersion")
#undef CN_SYM
  g_api = a;
  return &g_api;
}

#define CN_NEVENTS 64
struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;      // bucket all-reduces
  hipEvent_t ev[CN_NEVENTS] = {};   // zero-initialised: the clean-up path destroys only what was created
  int next_ev = 0;
  long long buckets = 0;
};

int hip_fail(const char* what, hipError_t e) {
  cn_set_error("cn_comm: %s: %s", what, hipGetErrorString(e));
  return CN_EHIP;
}
int rccl_fail(const RcclApi* api, const char* what, ncclResult_t r) {
  cn_set_error("cn_comm: %s: %s", what, api->GetErrorString(r));
  return CN_ERCCL;
}
#define CN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(#call, e_); } while (0)
#define CN_RCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return rccl_fail(api, #call, r_); } while (0)

hipEvent_t next_event(Comm* c) {
  hipEvent_t e = c->ev[c->next_ev];
  c->next_ev = (c->next_ev + 1) % CN_NEVENTS;
  return e;
}

// Releases whatever a (possibly partly initialised) communicator owns.
void comm_free(Comm* c) {
  if (c == nullptr) return;
  const RcclApi* api = rccl();
  if (c->stream != nullptr) (void)hipStreamSynchronize(c->stream);
  if (api != nullptr && c->comm != nullptr) (void)api->CommDestroy(c->comm);
  for (int i = 0; i < CN_NEVENTS; ++i)
    if (c->ev[i] != nullptr) (void)hipEventDestroy(c->ev[i]);
  if (c->stream != nullptr) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // namespace

// Resolves librccl and its entry points, nothing else (no communicator, no device work): phase 1 of the
// multi-rank set-up, so that a rank whose RCCL cannot be loaded is found out BEFORE any rank enters a
// collective (ncclCommInitRank blocks until every rank has arrived).
extern "C" int cn_comm_load(void) { return rccl() != nullptr ? CN_OK : CN_ERCCL; }

// Rank 0 calls this and ships the 128 bytes to every other rank.
extern "C" int cn_comm_unique_id(char* id128) {
  const RcclApi* api = rccl();
  if (api == nullptr) return CN_ERCCL;
  if (id128 == nullptr) { cn_set_error("cn_comm_unique_id: null buffer"); return CN_EINVAL; }
  ncclUniqueId id;
  CN_RCCL(api->GetUniqueId(&id));
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return CN_OK;
}

// Collective over all `world` ranks; binds the communicator to the CURRENT HIP device of the calling thread.
extern "C" int cn_comm_init(void** handle, const char* id128, int rank, int world) {
  const RcclApi* api = rccl();
  if (api == nullptr) return CN_ERCCL;
  if (handle == nullptr || id128 == nullptr || world < 1 || rank < 0 || rank >= world) {
    cn_set_error("cn_comm_init: bad arguments (rank %d of %d)", rank, world);
    return CN_EINVAL;
  }
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  // every failure below leaves through comm_free(): no leaked ncclComm_t / stream / events
#define CN_HIP_C(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { comm_free(c); return hip_fail(#call, e_); } } while (0)
  CN_HIP_C(hipGetDevice(&c->device));
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { c->comm = nullptr; comm_free(c); return rccl_fail(api, "ncclCommInitRank", r); }
  int lo = 0, hi = 0;
  CN_HIP_C(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CN_HIP_C(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));   // `hi` = numerically lowest = highest priority
  for (int i = 0; i < CN_NEVENTS; ++i) CN_HIP_C(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
#undef CN_HIP_C
  *handle = c;
  return CN_OK;
}

extern "C" int cn_comm_info(void* handle, int* rank, int* world, int* rccl_version) {
  Comm* c = (Comm*)handle;
  const RcclApi* api = rccl();
  if (c == nullptr || api == nullptr) { cn_set_error("cn_comm_info: no communicator"); return CN_EINVAL; }
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_version) CN_RCCL(api->GetVersion(rccl_version));
  return CN_OK;
}

// In-place SUM all-reduce of one gradient bucket (fp32) on the communicator's own stream, ordered after
// everything queued so far on the first `n_after` (0, 1 or 2) of the producer streams after_a, after_b
// (NULL there means the default stream, as everywhere in this ABI).
// (cn_comm_allreduce_bucket / _join / _allreduce: while a launch plan is being recorded - plan.hip - the call is
// logged, not issued: the recording runs under stream capture, RCCL runs live in every replay.)
void cn_plan_rec_comm(int kind, void* comm, void* buf, long long count, int dtype, void* s0, void* s1, int n_after);
int cn_comm_allreduce_bucket_issue(void* handle, float* buf, long long count, void* after_a, void* after_b, int n_after);
int cn_comm_join_issue(void* handle, void* stream);
int cn_comm_allreduce_issue(void* handle, void* buf, long long count, int dtype, void* stream);

extern "C" int cn_comm_allreduce_bucket(void* handle, float* buf, long long count, void* after_a, void* after_b,
                                        int n_after) {
  if (cn_plan_recording) {
    if (handle == nullptr || buf == nullptr || count <= 0 || n_after < 0 || n_after > 2) { cn_set_error("cn_comm_allreduce_bucket: bad arguments"); return CN_EINVAL; }
    cn_plan_rec_comm(0, handle, buf, count, 0, after_a, after_b, n_after);
    return CN_OK;
  }
  return cn_comm_allreduce_bucket_issue(handle, buf, count, after_a, after_b, n_after);
}
int cn_comm_allreduce_bucket_issue(void* handle, float* buf, long long count, void* after_a, void* after_b, int n_after) {
  Comm* c = (Comm*)handle;
  const RcclApi* api = rccl();
  if (c == nullptr || api == nullptr || buf == nullptr || count <= 0 || n_after < 0 || n_after > 2) {
    cn_set_error("cn_comm_allreduce_bucket: bad arguments");
    return CN_EINVAL;
  }
  hipStream_t prod[2] = {(hipStream_t)after_a, (hipStream_t)after_b};
  for (int i = 0; i < n_after; ++i) {
    if (i == 1 && prod[1] == prod[0]) continue;
    hipEvent_t e = next_event(c);
    CN_HIP(hipEventRecord(e, prod[i]));
    CN_HIP(hipStreamWaitEvent(c->stream, e, 0));
  }
  CN_RCCL(api->AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->comm, c->stream));
  c->buckets++;
  return CN_OK;
}

// `stream` waits for every bucket all-reduce queued so far.
extern "C" int cn_comm_join(void* handle, void* stream) {
  if (cn_plan_recording) {
    if (handle == nullptr) { cn_set_error("cn_comm_join: no communicator"); return CN_EINVAL; }
    cn_plan_rec_comm(1, handle, nullptr, 0, 0, stream, nullptr, 0);
    return CN_OK;
  }
  return cn_comm_join_issue(handle, stream);
}
int cn_comm_join_issue(void* handle, void* stream) {
  Comm* c = (Comm*)handle;
  if (c == nullptr) { cn_set_error("cn_comm_join: no communicator"); return CN_EINVAL; }
  hipEvent_t e = next_event(c);
  CN_HIP(hipEventRecord(e, c->stream));
  CN_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0));
  return CN_OK;
}

// In-stream, in-place SUM all-reduce.  dtype: 0 = fp32, 2 = fp64 (the SyncBatchNorm sums).
extern "C" int cn_comm_allreduce(void* handle, void* buf, long long count, int dtype, void* stream) {
  if (cn_plan_recording) {
    if (handle == nullptr || buf == nullptr || count <= 0 || (dtype != 0 && dtype != 2)) { cn_set_error("cn_comm_allreduce: bad arguments (dtype %d)", dtype); return CN_EINVAL; }
    cn_plan_rec_comm(2, handle, buf, count, dtype, stream, nullptr, 0);
    return CN_OK;
  }
  return cn_comm_allreduce_issue(handle, buf, count, dtype, stream);
}
int cn_comm_allreduce_issue(void* handle, void* buf, long long count, int dtype, void* stream) {
  Comm* c = (Comm*)handle;
  const RcclApi* api = rccl();
  if (c == nullptr || api == nullptr || buf == nullptr || count <= 0 || (dtype != 0 && dtype != 2)) {
    cn_set_error("cn_comm_allreduce: bad arguments (dtype %d)", dtype);
    return CN_EINVAL;
  }
  CN_RCCL(api->AllReduce(buf, buf, (size_t)count, dtype == 0 ? ncclFloat32 : ncclFloat64, ncclSum, c->comm,
                         (hipStream_t)stream));
  return CN_OK;
}

// In-stream broadcast of `nbytes` bytes from rank `root` (parameters at construction, BN buffers before validate).
extern "C" int cn_comm_broadcast(void* handle, void* buf, long long nbytes, int root, void* stream) {
  Comm* c = (Comm*)handle;
  const RcclApi* api = rccl();
  if (c == nullptr || api == nullptr || buf == nullptr || nbytes <= 0 || root < 0 || root >= c->world) {
    cn_set_error("cn_comm_broadcast: bad arguments");
    return CN_EINVAL;
  }
  CN_RCCL(api->Broadcast(buf, buf, (size_t)nbytes, ncclInt8, root, c->comm, (hipStream_t)stream));
  return CN_OK;
}

extern "C" int cn_comm_destroy(void* handle) {
  comm_free((Comm*)handle);
  return CN_OK;
}
#endif
