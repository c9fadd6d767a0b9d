// Library-owned RCCL communicator of a row-block sharded context (replaces the reference's MPI.COMM_WORLD,
// src/MultiGridBarrierMPI.jl:125,132, and the MPI layer of HPCSparseArrays).  The collectives of the Newton path are enqueued on
// the context stream -- no host synchronisation, no callback into the host language.  librccl is opened at run time
// (dlopen, local scope): the library still loads where RCCL is absent, and a host process that carries its own copy of RCCL
// (PyTorch does) keeps it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "amg.hpp"

namespace mgb {

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = std::getenv("MGB_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      if (!nm || !*nm) continue;
      api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
      api.error = dlerror();
    }
    if (!api.handle) return;
    auto sym = [&](const char* s) {
      void* p = dlsym(api.handle, s);
      if (!p) api.error = std::string("librccl lacks ") + s;
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
      dlclose(api.handle);
      api.handle = nullptr;
    }
  });
  return api;
}

RcclApi& need_rccl() {
  RcclApi& a = rccl();
  if (!a.handle) throw HipError("mgb: librccl could not be opened (" + a.error + ")");
  return a;
}

void rccl_check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw HipError(std::string("RCCL error in ") + what + ": " + need_rccl().GetErrorString(r));
}

}  // namespace

static_assert(NCCL_UNIQUE_ID_BYTES == 128, "mgb_rccl_unique_id hands out 128 bytes");

void rccl_unique_id(char* out128) {
  ncclUniqueId id;
  rccl_check(need_rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
}

void Ctx::set_comm_rccl(const char* id128, int rk, int wd) {
  if (wd < 1 || rk < 0 || rk >= wd || !id128) throw ArgError("ctx_set_comm_rccl: bad rank / world / id");
  RcclApi& a = need_rccl();
  hip_check(hipSetDevice(device), "hipSetDevice");
  drop_comm();
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  rccl_check(a.CommInitRank(&c, wd, id, rk), "ncclCommInitRank");
  rccl_comm = c;
  rank = rk;
  world = wd;
  allreduce = nullptr;
  allreduce_user = nullptr;
}

void Ctx::drop_comm() {
  if (rccl_comm) {
    (void)hipStreamSynchronize(stream);
    (void)rccl().CommDestroy((ncclComm_t)rccl_comm);
    rccl_comm = nullptr;
  }
}

void Ctx::rccl_allreduce(double* dev_ptr, long long count) {
  rccl_check(rccl().AllReduce(dev_ptr, dev_ptr, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)rccl_comm, stream), "ncclAllReduce");
}

}  // namespace mgb
