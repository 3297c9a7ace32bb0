// comm.hip -- data-parallel gradient exchange over RCCL/xGMI.
//
// The reference's multi-GPU path is one Python process that sums replica
// gradients onto GPU 0 (Link.addgrads, updaters.py:71-72), updates there, and
// copies parameters back (Link.copyparams, updaters.py:76-77).  Here every GPU is
// its own process; one in-place ncclAllReduce(sum, fp32) of the flat gradient
// arena replaces both steps (all ranks then apply the identical Adam update).
//
// librccl is dlopen()ed on first use so single-GPU runs never pay for loading it.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;

int load_rccl() {
  if (g_api.h) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) { vq::set_error("dlopen(librccl): %s", dlerror()); return VQVAE_E_NOLIB; }
  g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
  g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_api.CommCount = (decltype(g_api.CommCount))dlsym(h, "ncclCommCount");
  g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.AllReduce || !g_api.CommDestroy) {
    vq::set_error("librccl: missing symbols");
    dlclose(h);
    return VQVAE_E_NOLIB;
  }
  g_api.h = h;
  return 0;
}

int nccl_fail(const char* what, ncclResult_t r) {
  vq::set_error("%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error");
  return VQVAE_E_COMM;
}

}  // namespace

extern "C" {

int vqvae_comm_unique_id(char id[VQVAE_COMM_ID_BYTES]) {
  VQ_REQUIRE(id, "comm_unique_id: null");
  if (int e = load_rccl()) return e;
  static_assert(sizeof(ncclUniqueId) == VQVAE_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  ncclResult_t r = g_api.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  memcpy(id, &u, sizeof(u));
  return 0;
}

int vqvae_comm_init(void** comm, int nranks, int rank, const char id[VQVAE_COMM_ID_BYTES]) {
  VQ_REQUIRE(comm && id, "comm_init: null");
  VQ_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad rank %d/%d", rank, nranks);
  if (int e = load_rccl()) return e;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t c;
  ncclResult_t r = g_api.CommInitRank(&c, nranks, u, rank);
  if (r != ncclSuccess) return nccl_fail("ncclCommInitRank", r);
  *comm = (void*)c;
  return 0;
}

int vqvae_comm_allreduce_sum_f32(void* comm, float* buf, size_t n, vqvae_stream_t s) {
  VQ_REQUIRE(comm && buf, "comm_allreduce_sum: null");
  ncclResult_t r = g_api.AllReduce(buf, buf, n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)s);
  if (r != ncclSuccess) return nccl_fail("ncclAllReduce(sum)", r);
  return 0;
}

int vqvae_comm_allreduce_max_f32(void* comm, float* buf, size_t n, vqvae_stream_t s) {
  VQ_REQUIRE(comm && buf, "comm_allreduce_max: null");
  ncclResult_t r = g_api.AllReduce(buf, buf, n, ncclFloat32, ncclMax, (ncclComm_t)comm, (hipStream_t)s);
  if (r != ncclSuccess) return nccl_fail("ncclAllReduce(max)", r);
  return 0;
}

int vqvae_comm_count(void* comm, int* nranks) {
  VQ_REQUIRE(comm && nranks, "comm_count: null");
  VQ_REQUIRE(g_api.CommCount, "comm_count: librccl has no ncclCommCount");
  ncclResult_t r = g_api.CommCount((ncclComm_t)comm, nranks);
  if (r != ncclSuccess) return nccl_fail("ncclCommCount", r);
  return 0;
}

int vqvae_comm_destroy(void* comm) {
  if (!comm) return 0;
  ncclResult_t r = g_api.CommDestroy((ncclComm_t)comm);
  if (r != ncclSuccess) return nccl_fail("ncclCommDestroy", r);
  return 0;
}

}  // extern "C"
