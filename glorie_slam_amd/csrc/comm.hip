// The BA's one exchange step as a C-ABI entry point: a context-owned RCCL communicator and the
// all-reduce(sum) of the reduced normal equations [H | v] between glorie_ba_build_system and glorie_ba_solve_update.
// The reference has no counterpart (single GPU); the partition it serves is described at include/glorie_hip.h
// (glorie_ba_build_system).  RCCL is bound at first use with dlopen, so the library - and every single-GPU caller - has no
// load-time dependency on it; the collective runs on the caller's stream and can be recorded into a hipGraph with the
// launches around it (torch.distributed's all_reduce between two ctypes calls cannot).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>
#include "common.hiph"

namespace glorie {

struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  bool ok = false;
};

static Rccl& rccl() {
  static Rccl r = [] {
    Rccl t;
    // the copy that is already in the process (torch bundles one) before any other: two RCCL instances in one process
    // do not share their bootstrap state
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h && dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = dlopen(nullptr, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return t;
    t.get_unique_id = reinterpret_cast<decltype(t.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    t.comm_init_rank = reinterpret_cast<decltype(t.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    t.comm_destroy = reinterpret_cast<decltype(t.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    t.all_reduce = reinterpret_cast<decltype(t.all_reduce)>(dlsym(h, "ncclAllReduce"));
    t.all_gather = reinterpret_cast<decltype(t.all_gather)>(dlsym(h, "ncclAllGather"));
    t.ok = t.get_unique_id && t.comm_init_rank && t.comm_destroy && t.all_reduce && t.all_gather;
    return t;
  }();
  return r;
}

int comm_destroy(Ctx* ctx) {
  if (ctx->comm) {
    if (rccl().ok) (void)rccl().comm_destroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
  }
  return GLORIE_OK;
}

}  // namespace glorie

using namespace glorie;

static_assert(sizeof(ncclUniqueId) == GLORIE_COMM_ID_BYTES, "glorie_comm_unique_id hands out an ncclUniqueId");

extern "C" int glorie_comm_unique_id(void* id_out) {
  if (!id_out) return GLORIE_EINVAL;
  if (!rccl().ok) return GLORIE_EUNSUPPORTED;
  ncclUniqueId id;
  if (rccl().get_unique_id(&id) != ncclSuccess) return GLORIE_EHIP;
  memcpy(id_out, &id, sizeof(id));
  return GLORIE_OK;
}

extern "C" int glorie_comm_init(glorie_ctx* ctx, const void* id, int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return GLORIE_EINVAL;
  if (!rccl().ok) return GLORIE_EUNSUPPORTED;
  comm_destroy(ctx);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  if (rccl().comm_init_rank(&comm, world, uid, rank) != ncclSuccess) return GLORIE_EHIP;
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return GLORIE_OK;
}

extern "C" int glorie_comm_destroy(glorie_ctx* ctx) {
  if (!ctx) return GLORIE_EINVAL;
  return comm_destroy(ctx);
}

extern "C" int glorie_comm_world(const glorie_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_world : 0; }

extern "C" int glorie_allreduce_normal_eq(glorie_ctx* ctx, double* hv, size_t n, void* stream) {
  if (!ctx || (!hv && n)) return GLORIE_EINVAL;
  if (!ctx->comm) return GLORIE_EINVAL;                       // no communicator: the caller must not skip the exchange silently
  if (n == 0) return GLORIE_OK;
  if (rccl().all_reduce(hv, hv, n, ncclFloat64, ncclSum, static_cast<ncclComm_t>(ctx->comm),
                        reinterpret_cast<hipStream_t>(stream)) != ncclSuccess)
    return GLORIE_EHIP;
  return GLORIE_OK;
}

extern "C" int glorie_allgather_rows(glorie_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  if (!ctx || !ctx->comm || (!send && bytes_per_rank) || (!recv && bytes_per_rank)) return GLORIE_EINVAL;
  if (bytes_per_rank == 0) return GLORIE_OK;
  if (rccl().all_gather(send, recv, bytes_per_rank, ncclInt8, static_cast<ncclComm_t>(ctx->comm),
                        reinterpret_cast<hipStream_t>(stream)) != ncclSuccess)
    return GLORIE_EHIP;
  return GLORIE_OK;
}
