// Native collective: RCCL bound at run time (dlopen) so that hosts without torch.distributed -- the Julia glue, plain
// C clients -- get the row-shard all-reduce of SURVEY 8(e) from the library itself.  One communicator per context;
// the collective runs on a side stream behind an event of the context's stream, so the asynchronous pair overlaps the
// chunked pass T exactly like the torch.distributed callbacks do.
//
// No RCCL header is needed: the four entry points and the two enum values used are part of the stable NCCL ABI
// (ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0, 128-byte ncclUniqueId passed by value).
#include <dlfcn.h>

#include "pg_internal.h"

namespace {

struct NcclUniqueId {
  char internal[PG_COMM_ID_BYTES];
};
typedef int (*ncclGetUniqueId_t)(NcclUniqueId*);
typedef int (*ncclCommInitRank_t)(void**, int, NcclUniqueId, int);
typedef int (*ncclCommDestroy_t)(void*);
typedef int (*ncclAllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ncclGetErrorString_t)(int);

struct RcclApi {
  void* lib = nullptr;
  ncclGetUniqueId_t get_unique_id = nullptr;
  ncclCommInitRank_t comm_init_rank = nullptr;
  ncclCommDestroy_t comm_destroy = nullptr;
  ncclAllReduce_t all_reduce = nullptr;
  ncclGetErrorString_t error_string = nullptr;
};

RcclApi* rccl_api() {
  // (a function-local static: initialised once, thread-safe)
  static RcclApi api = [] {
    RcclApi a;
    // the copy already mapped by the process (torch ships one) wins; else the ROCm installation's
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (a.lib) {
      a.get_unique_id = (ncclGetUniqueId_t)dlsym(a.lib, "ncclGetUniqueId");
      a.comm_init_rank = (ncclCommInitRank_t)dlsym(a.lib, "ncclCommInitRank");
      a.comm_destroy = (ncclCommDestroy_t)dlsym(a.lib, "ncclCommDestroy");
      a.all_reduce = (ncclAllReduce_t)dlsym(a.lib, "ncclAllReduce");
      a.error_string = (ncclGetErrorString_t)dlsym(a.lib, "ncclGetErrorString");
    }
    return a;
  }();
  if (!api.lib || !api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce) return nullptr;
  return &api;
}

const char* rccl_err(RcclApi* api, int rc) { return api->error_string ? api->error_string(rc) : "?"; }

int native_allreduce(void* user, void* buf, int64_t count, int32_t dtype, void* stream) {
  pg_ctx* c = (pg_ctx*)user;
  RcclApi* api = rccl_api();
  c->comm->calls += 1;
  c->comm->elements += count;
  return api->all_reduce(buf, buf, (size_t)count, dtype == PG_F64 ? 8 : 7, 0, c->comm->comm, (hipStream_t)stream);
}

int native_allreduce_begin(void* user, void* buf, int64_t count, int32_t dtype, void* stream) {
  pg_ctx* c = (pg_ctx*)user;
  RcclApi* api = rccl_api();
  pg_comm* k = c->comm;
  k->calls += 1;
  k->elements += count;
  if (hipEventRecord(k->ev_ready, (hipStream_t)stream) != hipSuccess) return 1;
  if (hipStreamWaitEvent(k->side, k->ev_ready, 0) != hipSuccess) return 1;
  return api->all_reduce(buf, buf, (size_t)count, dtype == PG_F64 ? 8 : 7, 0, k->comm, k->side);
}

int native_allreduce_wait(void* user, void* stream) {
  pg_ctx* c = (pg_ctx*)user;
  pg_comm* k = c->comm;
  if (hipEventRecord(k->ev_done, k->side) != hipSuccess) return 1;
  if (hipStreamWaitEvent((hipStream_t)stream, k->ev_done, 0) != hipSuccess) return 1;
  return 0;
}

void install(pg_ctx* c, int32_t overlap) {
  c->allreduce = native_allreduce;
  c->allreduce_user = c;
  c->allreduce_begin = overlap ? native_allreduce_begin : nullptr;
  c->allreduce_wait = overlap ? native_allreduce_wait : nullptr;
}

}  // namespace

extern "C" {

int32_t pg_comm_available(void) { return rccl_api() != nullptr ? 1 : 0; }

pg_status pg_comm_get_unique_id(void* id_out) {
  PG_REQUIRE(id_out != nullptr, "id_out is null");
  RcclApi* api = rccl_api();
  if (!api) {
    pg_set_error("librccl could not be loaded (dlopen librccl.so.1)");
    return PG_ERR_UNSUPPORTED;
  }
  NcclUniqueId id;
  int rc = api->get_unique_id(&id);
  if (rc != 0) {
    pg_set_error("ncclGetUniqueId failed: %s", rccl_err(api, rc));
    return PG_ERR_COLLECTIVE;
  }
  memcpy(id_out, id.internal, PG_COMM_ID_BYTES);
  return PG_OK;
}

pg_status pg_ctx_comm_init(pg_ctx* c, const void* id_bytes, int32_t nranks, int32_t rank, int32_t overlap) {
  PG_REQUIRE(c != nullptr && id_bytes != nullptr, "null argument");
  PG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
  if (c->comm != nullptr) {
    // one communicator per context for the life of the job: a second call with the same (nranks, rank) re-installs the
    // library's all-reduce (a host callback may have replaced it meanwhile) and switches the overlap mode; the id is ignored
    PG_REQUIRE(c->comm->nranks == nranks && c->comm->rank == rank, "the context already has a communicator of another shape");
    install(c, overlap);
    return PG_OK;
  }
  RcclApi* api = rccl_api();
  if (!api) {
    pg_set_error("librccl could not be loaded (dlopen librccl.so.1)");
    return PG_ERR_UNSUPPORTED;
  }
  PG_HIP(hipSetDevice(c->device));
  pg_comm* k = new pg_comm();
  NcclUniqueId id;
  memcpy(id.internal, id_bytes, PG_COMM_ID_BYTES);
  int rc = api->comm_init_rank(&k->comm, nranks, id, rank);
  if (rc != 0) {
    pg_set_error("ncclCommInitRank(nranks=%d, rank=%d) failed: %s", nranks, rank, rccl_err(api, rc));
    delete k;
    return PG_ERR_COLLECTIVE;
  }
  if (hipStreamCreateWithFlags(&k->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&k->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&k->ev_done, hipEventDisableTiming) != hipSuccess) {
    pg_set_error("side stream / event creation failed");
    if (k->ev_done) (void)hipEventDestroy(k->ev_done);
    if (k->ev_ready) (void)hipEventDestroy(k->ev_ready);
    if (k->side) (void)hipStreamDestroy(k->side);
    api->comm_destroy(k->comm);
    delete k;
    return PG_ERR_HIP;
  }
  k->nranks = nranks;
  k->rank = rank;
  c->comm = k;
  install(c, overlap);
  return PG_OK;
}

pg_status pg_ctx_comm_stats(pg_ctx* c, int64_t* calls, int64_t* elements) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (calls) *calls = c->comm ? c->comm->calls : 0;
  if (elements) *elements = c->comm ? c->comm->elements : 0;
  return PG_OK;
}

pg_status pg_ctx_comm_destroy(pg_ctx* c) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (!c->comm) return PG_OK;
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->comm->side);
  RcclApi* api = rccl_api();
  if (api) api->comm_destroy(c->comm->comm);
  (void)hipEventDestroy(c->comm->ev_ready);
  (void)hipEventDestroy(c->comm->ev_done);
  (void)hipStreamDestroy(c->comm->side);
  delete c->comm;
  c->comm = nullptr;
  c->allreduce = nullptr;
  c->allreduce_begin = nullptr;
  c->allreduce_wait = nullptr;
  c->allreduce_user = nullptr;
  return PG_OK;
}

}  // extern "C"
