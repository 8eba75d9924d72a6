// Data-parallel gradient exchange behind the C-ABI (SURVEY.md 8(b) proposal: las_dp_init / allreduce_bucket / finalize): the
// cross-replica SUM that tf.tpu.CrossShardOptimizer performs in the reference (model_helper.py:405-406) as RCCL all-reduces over
// xGMI, for hosts that bind liblas_hip.so WITHOUT torch (phones-las_amd/dp.py keeps torch.distributed -- backend "nccl" IS RCCL on
// ROCm -- as the default transport of train.py / bench.py; both end in the same ncclAllReduce).
//
// RCCL is resolved at RUN time (dlopen), not linked: the library has no RCCL dependency unless a caller asks for a communicator,
// and a process that already holds an RCCL (torch's) gets that very copy (RTLD_NOLOAD first) -- two RCCL copies in one process
// would each start their own proxy threads and IPC state.  LAS_RCCL_LIB names another path.
#include "las_common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

namespace {

// the slice of rccl.h this file needs (ABI-stable across RCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };
enum { ncclSum = 0 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  const char* names[] = {getenv("LAS_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (int pass = 0; pass < 2 && !r.handle; ++pass)            // pass 0: a copy this process has already loaded
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (r.handle) break;
    }
  if (!r.handle) return nullptr;
#define LAS_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
  LAS_SYM(GetUniqueId, "ncclGetUniqueId");
  LAS_SYM(CommInitRank, "ncclCommInitRank");
  LAS_SYM(AllReduce, "ncclAllReduce");
  LAS_SYM(CommDestroy, "ncclCommDestroy");
  LAS_SYM(GetErrorString, "ncclGetErrorString");
  LAS_SYM(GroupStart, "ncclGroupStart");
  LAS_SYM(GroupEnd, "ncclGroupEnd");
#undef LAS_SYM
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { r.handle = nullptr; return nullptr; }
  return &r;
}

int check_nccl(Rccl* r, int rc, const char* what) {
  if (rc == ncclSuccess) return LAS_OK;
  las_set_error("%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
  return LAS_ERR_HIP;
}

}  // namespace

struct las_dp_comm {
  ncclComm_t comm;
  int rank, nranks;
};

extern "C" int las_dp_available(void) { return rccl() != nullptr ? 1 : 0; }

extern "C" int las_dp_unique_id(void* id_out) {
  LAS_REQUIRE(id_out != nullptr, "las_dp_unique_id: null argument");
  Rccl* r = rccl();
  LAS_REQUIRE(r != nullptr, "las_dp_unique_id: RCCL not found (librccl.so; LAS_RCCL_LIB names another path)");
  ncclUniqueId id;
  int rc = check_nccl(r, r->GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(id_out, &id, sizeof id);
  return LAS_OK;
}

extern "C" int las_dp_init(const void* id, int rank, int nranks, las_dp_comm** comm_out) {
  LAS_REQUIRE(id != nullptr && comm_out != nullptr && nranks > 0 && rank >= 0 && rank < nranks, "las_dp_init: bad arguments (rank %d of %d)", rank, nranks);
  Rccl* r = rccl();
  LAS_REQUIRE(r != nullptr, "las_dp_init: RCCL not found (librccl.so; LAS_RCCL_LIB names another path)");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t c = nullptr;
  int rc = check_nccl(r, r->CommInitRank(&c, nranks, uid, rank), "ncclCommInitRank");      // (uses the calling thread's current HIP device)
  if (rc) return rc;
  las_dp_comm* out = new las_dp_comm{c, rank, nranks};
  *comm_out = out;
  return LAS_OK;
}

extern "C" int las_dp_allreduce_bucket(las_dp_comm* comm, float* grads, int64_t count, void* stream) {
  LAS_REQUIRE(comm != nullptr && grads != nullptr && count > 0, "las_dp_allreduce_bucket: bad arguments");
  Rccl* r = rccl();
  LAS_REQUIRE(r != nullptr, "las_dp_allreduce_bucket: RCCL not loaded");
  return check_nccl(r, r->AllReduce(grads, grads, (size_t)count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int las_dp_finalize(las_dp_comm* comm) {
  if (comm == nullptr) return LAS_OK;
  Rccl* r = rccl();
  int rc = LAS_OK;
  if (r != nullptr && comm->comm != nullptr) rc = check_nccl(r, r->CommDestroy(comm->comm), "ncclCommDestroy");
  delete comm;
  return rc;
}
