// glh_comm.h -- the one collective of a multi-GPU run, native: RCCL over xGMI behind the C ABI.
//
// Tracked points are independent (tracker.py:305-374 closes only over read-only images / cameras; the
// reference's own parallelism is a map over tracks, tracker.py:381-387, helpers.py:2008-2017), so one
// process per GPU tracks a contiguous block of points and NOTHING is exchanged while a sequence runs.  At
// the end every rank's posterior moments [T][P_rank][12] (+ the per-point status word) go to the root with
// ONE grouped ncclSend / ncclRecv exchange: a gather with ragged blocks.  xGMI is point to point, every
// sender has its own link to the root, so the exchange is bound by the root's links, not by a ring.
//
// librccl is loaded with dlopen when the first communicator is made: a single-GPU user of the library never
// maps it.  The 128-byte unique id travels between the processes by whatever the launcher has (a file, a TCP
// store: glimpse_amd/sharding.py).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

namespace glh {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;  // why loading failed
};

// Loads librccl once per process.  Returns null (and leaves `error`) when it is not there.
inline RcclApi g_rccl;
inline const char* rccl_load_error() { return g_rccl.error.empty() ? "librccl is not available" : g_rccl.error.c_str(); }
inline RcclApi* rccl_api() {
  RcclApi& api = g_rccl;
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
  }
  if (!api.handle) {
    const char* e = dlerror();
    api.error = std::string("librccl could not be loaded: ") + (e ? e : "unknown error");
    return nullptr;
  }
  bool ok = true;
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(api.handle, name);
    if (!p) {
      ok = false;
      api.error = std::string("librccl has no symbol ") + name;
    }
    return p;
  };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!ok) {
    dlclose(api.handle);
    api.handle = nullptr;
    return nullptr;
  }
  return &api;
}

// A context's communicator: rank / world, the RCCL handle and the root's receive staging.
struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  double* stage = nullptr;   // root: [sum_r n_frames * P_r * 12] doubles, then [sum_r P_r] status words
  size_t stage_bytes = 0;
  size_t gathered_doubles = 0, gathered_points = 0, gathered_status_off = 0;  // what the last gather left in `stage`
  double* scalar = nullptr;  // one double / one int for the barrier and the max-reduction
};

}  // namespace glh
