// RCCL wrappers of the C ABI ("sp_allreduce_flat (RCCL wrapper taking an ncclComm_t created by the host side)").
//
// The path has ONE data-path collective: the sum of the flat fp32 gradient buffer over the data-parallel replicas
// (learner/Learner.py:120-122 run per replica).  torch.distributed can do it (parallel.py, the default); these entry
// points put the same collective on a stream the CALLER chooses -- a side stream forked inside a captured training step, or the
// two halves of a two-shot all-reduce (reduce-scatter + all-gather: every one of the 7 xGMI links of a GPU carries 1/8 of the
// buffer at once instead of the ring's neighbour-to-neighbour hops) -- without a torch type in the signature.
//
// librccl is resolved at run time (dlopen of the copy the process already has: torch ships its own), so the library loads on
// boxes without RCCL and the kernels' C ABI does not link against it.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <mutex>

#include "sp_common.h"

namespace {
typedef struct { char internal[128]; } sp_nccl_uid;            // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_uid)(sp_nccl_uid*);
typedef int (*fn_init_rank)(void**, int, sp_nccl_uid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_reduce_scatter reduce_scatter = nullptr;
  fn_allgather allgather = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
const int kNcclFloat = 7, kNcclDouble = 8, kNcclSum = 0;       // ncclFloat32, ncclFloat64, ncclSum (rccl.h)

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);              // the copy this process already loaded (torch's), if any
    if (g_rccl.h) break;
  }
  for (int i = 0; !g_rccl.h && i < 2; ++i) g_rccl.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.h) return;
  g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.h, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
  g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
  g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.h, "ncclAllReduce");
  g_rccl.reduce_scatter = (fn_reduce_scatter)dlsym(g_rccl.h, "ncclReduceScatter");
  g_rccl.allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather");
  g_rccl.errstr = (fn_errstr)dlsym(g_rccl.h, "ncclGetErrorString");
}
int need_rccl(const char* what) {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.h || !g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allreduce || !g_rccl.reduce_scatter || !g_rccl.allgather) {
    const char* why = dlerror();      // (one call: dlerror() clears the pending message)
    sp_set_error("%s: librccl.so is not available in this process (%s)", what, why ? why : "symbols missing");
    return SP_EHIP;
  }
  return SP_OK;
}
int check(int rc, const char* what) {
  if (rc == 0) return SP_OK;
  sp_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
  return SP_EHIP;
}
}  // namespace

extern "C" int sp_comm_available(void) { std::call_once(g_once, load_rccl); return g_rccl.h && g_rccl.allreduce ? 1 : 0; }

extern "C" int sp_comm_unique_id(void* id128) {
  SP_CHECK_ARG(id128, "sp_comm_unique_id: null pointer");
  if (int rc = need_rccl("sp_comm_unique_id")) return rc;
  sp_nccl_uid u;
  if (int rc = check(g_rccl.get_uid(&u), "sp_comm_unique_id")) return rc;
  memcpy(id128, &u, sizeof u);
  return SP_OK;
}

extern "C" int sp_comm_init_rank(void** comm, int32_t nranks, const void* id128, int32_t rank) {
  SP_CHECK_ARG(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "sp_comm_init_rank: bad arguments");
  if (int rc = need_rccl("sp_comm_init_rank")) return rc;
  sp_nccl_uid u;
  memcpy(&u, id128, sizeof u);
  return check(g_rccl.init_rank(comm, nranks, u, rank), "sp_comm_init_rank");
}

extern "C" int sp_comm_destroy(void* comm) {
  if (!comm) return SP_OK;
  if (int rc = need_rccl("sp_comm_destroy")) return rc;
  return check(g_rccl.destroy(comm), "sp_comm_destroy");
}

extern "C" int sp_allreduce_flat(void* comm, float* buf, int64_t n, sp_stream_t stream) {
  SP_CHECK_ARG(comm && buf && n > 0, "sp_allreduce_flat: bad arguments");
  if (int rc = need_rccl("sp_allreduce_flat")) return rc;
  return check(g_rccl.allreduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, comm, reinterpret_cast<hipStream_t>(stream)), "sp_allreduce_flat");
}

// the fp64 accumulators of the exact data-parallel mode (BatchNorm sums, Dice sums: a few KB each, include/stroke_amd.h conventions)
extern "C" int sp_allreduce_flat_f64(void* comm, double* buf, int64_t n, sp_stream_t stream) {
  SP_CHECK_ARG(comm && buf && n > 0, "sp_allreduce_flat_f64: bad arguments");
  if (int rc = need_rccl("sp_allreduce_flat_f64")) return rc;
  return check(g_rccl.allreduce(buf, buf, (size_t)n, kNcclDouble, kNcclSum, comm, reinterpret_cast<hipStream_t>(stream)), "sp_allreduce_flat_f64");
}

// two-shot form: rank r ends with the sum of elements [r*chunk, (r+1)*chunk) after the reduce-scatter (in place, at its own
// offset) and with the whole sum after the all-gather; n must be nranks * chunk (the caller pads the flat buffer)
extern "C" int sp_reduce_scatter_flat(void* comm, float* buf, int64_t chunk, int32_t rank, sp_stream_t stream) {
  SP_CHECK_ARG(comm && buf && chunk > 0 && rank >= 0, "sp_reduce_scatter_flat: bad arguments");
  if (int rc = need_rccl("sp_reduce_scatter_flat")) return rc;
  return check(g_rccl.reduce_scatter(buf, buf + (size_t)rank * chunk, (size_t)chunk, kNcclFloat, kNcclSum, comm, reinterpret_cast<hipStream_t>(stream)),
               "sp_reduce_scatter_flat");
}
extern "C" int sp_allgather_flat(void* comm, float* buf, int64_t chunk, int32_t rank, sp_stream_t stream) {
  SP_CHECK_ARG(comm && buf && chunk > 0 && rank >= 0, "sp_allgather_flat: bad arguments");
  if (int rc = need_rccl("sp_allgather_flat")) return rc;
  return check(g_rccl.allgather(buf + (size_t)rank * chunk, buf, (size_t)chunk, kNcclFloat, comm, reinterpret_cast<hipStream_t>(stream)), "sp_allgather_flat");
}
