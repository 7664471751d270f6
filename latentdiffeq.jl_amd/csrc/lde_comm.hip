// lde_comm.hip — the one collective of the path behind the C ABI: an in-place f32 sum-all-reduce over RCCL.
//
// The reference has no multi-process code (its only parallelism is EnsembleThreads() over trajectories
// [REF src/models/GOKU.jl:121]); trajectories are independent [REF GOKU.jl:111], so the batch shards by columns with no
// collective inside the solve, and the single exchange is the sum of the SHARED parameters' gradients (the RHS-MLP dW of
// lde_adjoint, plus a trainer's encoder / decoder gradients) once per optimiser step [REF examples/pendulum_friction-less/
// model_train.jl:186-204 is the step it sits in]. A Julia host that reaches the solver through `ccall` needs that exchange
// through the same boundary: one process per GPU, one communicator per process, xGMI underneath.
//
// librccl is bound at run time (dlopen), preferring a copy that is already in the process (torch ships its own; two RCCL
// copies would each bring their own proxy threads), so liblde.so itself carries no link-time dependency on it and loads
// on a box without RCCL — lde_comm_* then return LDE_ERR_UNSUPPORTED.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only; every function is looked up with dlsym

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/lde.h"

static_assert(LDE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "lde.h's id size must be RCCL's");

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("LDE_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so"};
    // a copy already mapped into the process first (RTLD_NOLOAD), then a fresh load
    for (int pass = 0; pass < 2 && !r.so; pass++)
      for (const char* n : names) {
        if (!n || !*n) continue;
        r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (r.so) break;
      }
    if (!r.so) {
      const char* e = dlerror();
      r.why = std::string("librccl not found: ") + (e ? e : "dlopen failed");
      return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    if (!r.ok) r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
  });
  return r;
}

std::string g_err;   // errors before a communicator exists

std::string nccl_text(const char* what, ncclResult_t rc) {
  Rccl& r = rccl();
  return std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string((int)rc) + ")";
}

}  // namespace

struct lde_comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
  std::string err;
};

extern "C" {

int lde_comm_unique_id(char* id) {
  if (!id) return LDE_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) {
    g_err = r.why;
    return LDE_ERR_UNSUPPORTED;
  }
  ncclUniqueId u;
  const ncclResult_t rc = r.GetUniqueId(&u);
  if (rc != ncclSuccess) {
    g_err = nccl_text("ncclGetUniqueId", rc);
    return LDE_ERR_HIP;
  }
  std::memcpy(id, u.internal, LDE_COMM_ID_BYTES);
  return LDE_OK;
}

int lde_comm_init(lde_comm** out, int nranks, int rank, const char* id) {
  if (!out) return LDE_ERR_INVALID_ARG;
  *out = nullptr;
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) {
    g_err = "lde_comm_init: need 0 <= rank < nranks and a unique id";
    return LDE_ERR_INVALID_ARG;
  }
  Rccl& r = rccl();
  if (!r.ok) {
    g_err = r.why;
    return LDE_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_err = "lde_comm_init: no HIP device";
    return LDE_ERR_NO_DEVICE;
  }
  lde_comm* c = new (std::nothrow) lde_comm();
  if (!c) return LDE_ERR_ALLOC;
  c->nranks = nranks;
  c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return LDE_ERR_NO_DEVICE;
  }
  ncclUniqueId u;
  std::memcpy(u.internal, id, LDE_COMM_ID_BYTES);
  const ncclResult_t rc = r.CommInitRank(&c->comm, nranks, u, rank);   // collective: every rank of the job calls it
  if (rc != ncclSuccess) {
    g_err = nccl_text("ncclCommInitRank", rc);
    delete c;
    return LDE_ERR_HIP;
  }
  *out = c;
  return LDE_OK;
}

int lde_comm_allreduce_f32(lde_comm* c, float* buf, int64_t n, void* stream) {
  if (!c) return LDE_ERR_INVALID_ARG;
  if (n < 0 || (n && !buf)) {
    c->err = "lde_comm_allreduce_f32: NULL buffer";
    return LDE_ERR_INVALID_ARG;
  }
  if (!n) return LDE_OK;
  const ncclResult_t rc = rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
  if (rc != ncclSuccess) {
    c->err = nccl_text("ncclAllReduce", rc);
    return LDE_ERR_HIP;
  }
  return LDE_OK;
}

int lde_comm_nranks(const lde_comm* c) { return c ? c->nranks : 0; }
int lde_comm_rank(const lde_comm* c) { return c ? c->rank : -1; }

void lde_comm_destroy(lde_comm* c) {
  if (!c) return;
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
}

const char* lde_comm_last_error(const lde_comm* c) { return c ? c->err.c_str() : g_err.c_str(); }

}  // extern "C"
