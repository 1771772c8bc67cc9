// nasr_comm.hip — the in-library gradient exchange: average_gradients (networks/tfnetwork.py:72-86) for hosts without
// torch.distributed.  librccl is bound at run time (dlopen), only when a host asks for it: a single-GPU host never loads it.
#include <dlfcn.h>

#include "nasr_ctx.h"

using namespace nasr;
using namespace nasr_impl;

namespace {
struct RcclApi {
  struct Uid { char internal[128]; };
  int (*GetUniqueId)(Uid*) = nullptr;
  int (*CommInitRank)(void**, int, Uid, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommSplit)(void*, int, int, void**, void*) = nullptr;     // optional (RCCL >= 2.18): a second communicator of the same ranks
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};
RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  void* lib = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) {
    api.why = std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : "?");
    return api;
  }
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  api.CommSplit = reinterpret_cast<decltype(api.CommSplit)>(dlsym(lib, "ncclCommSplit"));
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
  if (!api.ok) api.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
  return api;
}
constexpr int kNcclFloat = 7, kNcclSum = 0;      // rccl.h: ncclFloat32, ncclSum
int rccl_fail(nasr_ctx* h, const char* what, int rc) {
  const RcclApi& r = rccl();
  return h->fail(NASR_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" +
                                   std::to_string(rc) + ")");
}
}  // namespace

extern "C" {

int nasr_comm_unique_id(void* id128) {
  if (!id128) return NASR_ERR_ARG;
  RcclApi& r = rccl();
  if (!r.ok) {
    g_create_error = r.why;
    return NASR_ERR_HIP;
  }
  RcclApi::Uid u;
  const int rc = r.GetUniqueId(&u);
  if (rc) {
    g_create_error = std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return NASR_ERR_HIP;
  }
  memcpy(id128, u.internal, 128);
  return NASR_OK;
}

int nasr_comm_init(nasr_handle h, const void* id128, int rank, int nranks) {
  if (!h || !id128) return NASR_ERR_ARG;
  if (nranks < 1 || rank < 0 || rank >= nranks) return h->fail(NASR_ERR_ARG, "nasr_comm_init: bad rank / nranks");
  if (h->comm) return h->fail(NASR_ERR_STATE, "nasr_comm_init: this handle already has a communicator");
  RcclApi& r = rccl();
  if (!r.ok) return h->fail(NASR_ERR_HIP, r.why);
  HIPCHK(h, hipSetDevice(h->device));
  RcclApi::Uid u;
  memcpy(u.internal, id128, 128);
  void* c = nullptr;
  const int rc = r.CommInitRank(&c, nranks, u, rank);       // blocks until every rank has joined
  if (rc) return rccl_fail(h, "ncclCommInitRank", rc);
  h->comm = c;
  h->comm_rank = rank;
  h->comm_n = nranks;
  HIPCHK(h, hipStreamCreateWithFlags(&h->comm_st, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
  HIPCHK(h, hipMalloc(&h->comm_scratch, 64 * sizeof(float)));
  if (r.CommSplit) {                      // collective over all ranks of `comm`: every rank gets here (same library everywhere)
    void* c2 = nullptr;
    if (r.CommSplit(c, 0, rank, &c2, nullptr) == 0 && c2) {
      h->comm2 = c2;
      HIPCHK(h, hipStreamCreateWithFlags(&h->comm_st2, hipStreamNonBlocking));
    }
  }
  return NASR_OK;
}

int nasr_comm_size(nasr_handle h) { return h ? (h->comm ? h->comm_n : 1) : NASR_ERR_ARG; }

int nasr_comm_allreduce_grads(nasr_handle h) {
  if (!h) return NASR_ERR_ARG;
  if (!h->comm) return h->fail(NASR_ERR_STATE, "nasr_comm_allreduce_grads: call nasr_comm_init first");
  if (!h->have_grads) return h->fail(NASR_ERR_STATE, "nasr_comm_allreduce_grads without gradients");
  RcclApi& r = rccl();
  HIPCHK(h, hipSetDevice(h->device));
  // bucket i crosses xGMI as soon as the backward pass has finished it (its event, held back over the next persistent
  // BPTT launch when bucket_defer is on), under the layers below; the handle's stream then waits for the last collective
  for (size_t i = 0; i < h->buckets.size(); ++i) {
    HIPCHK(h, hipStreamWaitEvent(h->comm_st, h->ev_bucket[i], 0));
    float* p = h->Gbase + h->buckets[i].first;
    const int rc = r.AllReduce(p, p, (size_t)h->buckets[i].second, kNcclFloat, kNcclSum, h->comm, h->comm_st);
    if (rc) return rccl_fail(h, "ncclAllReduce", rc);
  }
  HIPCHK(h, hipEventRecord(h->ev_comm, h->comm_st));
  HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_comm, 0));
  return NASR_OK;
}

int nasr_comm_mean(nasr_handle h, float* vals, int n) {
  if (!h || !vals) return NASR_ERR_ARG;
  if (n < 1 || n > 64) return h->fail(NASR_ERR_ARG, "nasr_comm_mean: 1..64 values");
  if (!h->comm) return NASR_OK;                   // one rank: the mean is the value
  RcclApi& r = rccl();
  HIPCHK(h, hipSetDevice(h->device));
  // On its own communicator and stream when the library offers ncclCommSplit: the collective of a few floats neither waits
  // for the gradient buckets of the step in flight nor for the compute stream.  Otherwise (one communicator executes its
  // collectives in issue order) it goes behind them on the compute stream, as documented in include/nasr.h.
  void* c = h->comm2 ? h->comm2 : h->comm;
  hipStream_t st = h->comm2 ? h->comm_st2 : h->st;
  HIPCHK(h, hipMemcpyAsync(h->comm_scratch, vals, (size_t)n * 4, hipMemcpyHostToDevice, st));
  const int rc = r.AllReduce(h->comm_scratch, h->comm_scratch, (size_t)n, kNcclFloat, kNcclSum, c, st);
  if (rc) return rccl_fail(h, "ncclAllReduce", rc);
  HIPCHK(h, hipMemcpyAsync(vals, h->comm_scratch, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) vals[i] /= (float)h->comm_n;
  return NASR_OK;
}

int nasr_comm_destroy(nasr_handle h) {
  if (!h) return NASR_ERR_ARG;
  if (!h->comm) return NASR_OK;
  (void)hipSetDevice(h->device);
  if (h->comm_st) (void)hipStreamSynchronize(h->comm_st);
  if (h->comm_st2) (void)hipStreamSynchronize(h->comm_st2);
  if (h->comm2) (void)rccl().CommDestroy(h->comm2);
  h->comm2 = nullptr;
  if (h->comm_st2) (void)hipStreamDestroy(h->comm_st2);
  h->comm_st2 = nullptr;
  (void)rccl().CommDestroy(h->comm);
  h->comm = nullptr;
  if (h->comm_st) (void)hipStreamDestroy(h->comm_st);
  if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
  if (h->comm_scratch) (void)hipFree(h->comm_scratch);
  h->comm_st = nullptr; h->ev_comm = nullptr; h->comm_scratch = nullptr;
  h->comm_n = 1; h->comm_rank = 0;
  return NASR_OK;
}
}  // extern "C"
