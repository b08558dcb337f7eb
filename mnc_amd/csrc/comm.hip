// Multi-GPU exchange of the batched-image path: images are sharded one per rank with no data-path collective;
// the only exchange is an all-gather of each rank's fixed-shape block of final instances ([100][447] float32, 179 KB) over
// RCCL / xGMI -- latency-bound, far below one link's bandwidth.  The collective is issued directly on the context's stream
// (the stream the voting kernels wrote the block on), device pointer in, device pointer out: no host hop, no second stream,
// no framework tensor.
//
// librccl is loaded at run time (dlopen) so that libmnc_hip.so has no link-time dependency on it: single-GPU users
// never touch it, and inside a process that already loaded an RCCL (e.g. torch.distributed's) the same library instance is
// reused.  The 128-byte ncclUniqueId is created by rank 0 (mnc_comm_unique_id) and carried to the other ranks by whatever
// rendezvous the host program has (bench.py: torch.distributed's store; a C host: a file, MPI, a socket).
#include <dlfcn.h>

#include "mnc_internal.h"

namespace mnc {

typedef int ncclResult_t;                      // rccl.h: ncclSuccess == 0
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };   // rccl.h: NCCL_UNIQUE_ID_BYTES == 128
enum { kNcclFloat32 = 7 };                     // rccl.h: ncclFloat32 == ncclFloat == 7

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.handle) return MNC_OK;
  // The RCCL that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm installations (a torch
  // wheel bundles its own libamdhip64 / librccl with the same sonames as /opt/rocm's), and a librccl resolved by soname could
  // be the one attached to the other runtime -- whose communicators cannot see this library's devices or streams.  So look next
  // to the libamdhip64 that hipGetDeviceCount resolves to first, by absolute path, and only then by soname.
  void* h = nullptr;
  Dl_info info;
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      for (const char* leaf : {"librccl.so.1", "librccl.so"}) {
        h = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    }
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (h) break;
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) {
    set_error("RCCL is not available: %s", dlerror());
    return MNC_ERR_UNSUPPORTED;
  }
  Rccl r;
  r.handle = h;
  r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetVersion || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
    set_error("librccl lacks an expected symbol");
    return MNC_ERR_UNSUPPORTED;
  }
  g_rccl = r;
  return MNC_OK;
}

#define MNC_NCCL_TRY(expr)                                                                      \
  do {                                                                                          \
    ncclResult_t r__ = (expr);                                                                  \
    if (r__ != 0) {                                                                             \
      set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r__));                            \
      return MNC_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)

struct CommState {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
};

void comm_free(mnc_ctx* ctx) {
  CommState* st = (CommState*)ctx->comm;
  if (!st) return;
  if (st->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(st->comm);
  delete st;
  ctx->comm = nullptr;
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_comm_unique_id(void* id_out, int capacity_bytes) {
  MNC_REQUIRE(id_out && capacity_bytes >= (int)sizeof(ncclUniqueId), "mnc_comm_unique_id: need a %zu-byte buffer",
              sizeof(ncclUniqueId));
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  MNC_NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  clear_error();
  return MNC_OK;
}

int mnc_comm_init(mnc_ctx* ctx, const void* id, int nranks, int rank) {
  MNC_REQUIRE(ctx && id && nranks >= 1 && rank >= 0 && rank < nranks, "mnc_comm_init: bad argument (nranks=%d rank=%d)", nranks,
              rank);
  MNC_REQUIRE(!ctx->comm, "mnc_comm_init: this context already has a communicator");
  int rc = rccl_load();
  if (rc) return rc;
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  CommState* st = new CommState();
  st->nranks = nranks;
  st->rank = rank;
  ncclResult_t r = g_rccl.CommInitRank(&st->comm, nranks, uid, rank);
  if (r != 0) {
    set_error("ncclCommInitRank(nranks=%d, rank=%d, device %d) failed: %s", nranks, rank, ctx->device, g_rccl.GetErrorString(r));
    delete st;
    return MNC_ERR_HIP;
  }
  ctx->comm = st;
  clear_error();
  return MNC_OK;
}

int mnc_comm_info(mnc_ctx* ctx, int* nranks, int* rank, int* rccl_version) {
  MNC_REQUIRE(ctx, "mnc_comm_info: null context");
  CommState* st = (CommState*)ctx->comm;
  if (nranks) *nranks = st ? st->nranks : 0;
  if (rank) *rank = st ? st->rank : 0;
  if (rccl_version) {
    *rccl_version = 0;
    if (rccl_load() == MNC_OK) (void)g_rccl.GetVersion(rccl_version);
  }
  clear_error();
  return MNC_OK;
}

int mnc_gather_instances(mnc_ctx* ctx, const float* d_send, float* d_recv, size_t floats_per_rank) {
  MNC_REQUIRE(ctx && d_send && d_recv && floats_per_rank > 0, "mnc_gather_instances: bad argument");
  CommState* st = (CommState*)ctx->comm;
  MNC_REQUIRE(st && st->comm, "mnc_gather_instances: mnc_comm_init has not run on this context");
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  LaunchScope ls(ctx, "rccl_all_gather", 0.0, (double)floats_per_rank * 4.0 * st->nranks);
  MNC_NCCL_TRY(g_rccl.AllGather(d_send, d_recv, floats_per_rank, kNcclFloat32, st->comm, ctx->stream));
  return ls.finish("ncclAllGather");
}

int mnc_comm_destroy(mnc_ctx* ctx) {
  MNC_REQUIRE(ctx, "mnc_comm_destroy: null context");
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    comm_free(ctx);
  }
  clear_error();
  return MNC_OK;
}

}  // extern "C"
