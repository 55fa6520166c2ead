// povar_comm.hip -- exchange steps of the sharded path: RCCL / host-hook all-reduce, the peer-to-peer term exchange.
#include "povar_ctx.hpp"

int allreduce(povar_ctx* c, double* buf, size_t n) {
  if (c->host_fn) {
    prof_mark(c, 2);
    c->host_stage.resize(n);
    HIP_TRY(hipMemcpyAsync(c->host_stage.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->host_fn(c->host_stage.data(), (int64_t)n, c->host_user);
    HIP_TRY(hipMemcpyAsync(buf, c->host_stage.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (!c->comm) return 0;
  prof_mark(c, 2);
  NCCL_TRY(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, c->comm, c->stream));
  return 0;
}

// peer-to-peer fields of the term kernels (only while the exchange is attached)
void p2p_dp(povar_ctx* c, Dp& dt) {
  if (!c->p2p) return;
  dt.p2p_peer = c->peer_dev.p;
  dt.p2p_epoch = c->p2p_epoch.p;
  dt.p2p_world = c->world;
  dt.p2p_rank = c->rank;
}

extern "C" {

int povar_comm_ranks(povar_ctx* c) {
  if (!c) return fail(-1, "null context");
  if (c->host_fn) return c->world;
  if (!c->comm) return 0;
  int n = 0;
  NCCL_TRY(ncclCommCount(c->comm, &n));
  return n;
}

int povar_p2p_export(povar_ctx* c, int32_t world, uint8_t handle[64]) {
  if (int rc = check_ctx(c)) return rc;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
  if (world < 1 || !handle) return fail(-1, "bad p2p arguments");
  if (!c->xbuf) {
    c->xbuf_count = (size_t)2 * world * c->n_cams * 16;
    // fine-grained device memory: peers' stores and this GPU's system-scope loads meet in memory, not in an L2
    hipError_t e = hipExtMallocWithFlags((void**)&c->xbuf, c->xbuf_count * sizeof(double), hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      HIP_TRY(hipMalloc((void**)&c->xbuf, c->xbuf_count * sizeof(double)));
    }
    c->bytes += c->xbuf_count * sizeof(double);
    HIP_TRY(hipMemset(c->xbuf, 0xff, c->xbuf_count * sizeof(double)));  // tags != any epoch
    HIP_TRY(hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, c->xbuf));
  std::memcpy(handle, &h, 64);
  return 0;
}

int povar_p2p_attach(povar_ctx* c, int32_t world, int32_t rank, const uint8_t* handles) {
  if (int rc = check_ctx(c)) return rc;
  if (world < 1 || rank < 0 || rank >= world || !handles || !c->xbuf) return fail(-1, "bad p2p arguments (export first)");
  if ((size_t)2 * world * c->n_cams * 16 != c->xbuf_count) return fail(-1, "p2p world size differs from the exported buffer");
  // the once-per-solve exchanges (G, b, scalars) stay on the communicator: the push/reduce kernels only replace the
  // per-term all-reduce of an already sharded context
  if (!(c->comm || c->host_fn) || c->world != world || c->rank != rank)
    return fail(-1, "povar_p2p_attach needs the communicator of the same world/rank attached first (povar_comm_init)");
  // re-attach: drop the mappings, the pointer table and the captured term loop of the previous attachment
  for (size_t p = 0; p < c->peer_host.size(); ++p)
    if (c->peer_host[p] && c->peer_host[p] != c->xbuf) (void)hipIpcCloseMemHandle(c->peer_host[p]);
  c->peer_host.clear();
  c->peer_dev.release();
  c->p2p_epoch.release();
  if (c->series_graph) { (void)hipGraphExecDestroy(c->series_graph); c->series_graph = nullptr; }
  c->p2p = false;
  c->peer_host.assign(world, nullptr);
  for (int p = 0; p < world; ++p) {
    if (p == rank) { c->peer_host[p] = c->xbuf; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + 64 * (size_t)p, 64);
    void* ptr = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    c->peer_host[p] = (double*)ptr;
  }
  HIP_TRY(c->peer_dev.alloc(world, &c->bytes));
  HIP_TRY(hipMemcpy(c->peer_dev.p, c->peer_host.data(), world * sizeof(double*), hipMemcpyHostToDevice));
  HIP_TRY(c->p2p_epoch.alloc(1, &c->bytes));
  HIP_TRY(hipMemset(c->p2p_epoch.p, 0, sizeof(unsigned long long)));
  HIP_TRY(hipDeviceSynchronize());
  c->world = world;
  c->rank = rank;
  c->p2p = true;
  if (!c->lpl_forced) c->use_lpl = true;  // the push/reduce exchange belongs to the lane-per-landmark term kernels
  return 0;
}

int povar_p2p_enable(povar_ctx* c, int32_t on) {
  if (int rc = check_ctx(c)) return rc;
  if (on && !c->peer_dev.p) return fail(-1, "povar_p2p_enable before povar_p2p_attach");
  c->p2p = on != 0;
  return 0;
}

int povar_comm_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  std::memcpy(id, &u, 128);
  return 0;
}

int povar_comm_init_host(povar_ctx* c, int32_t world, int32_t rank, povar_allreduce_fn fn, void* user) {
  if (int rc = check_ctx(c)) return rc;
  if (world < 1 || rank < 0 || rank >= world || !fn) return fail(-1, "bad communicator arguments");
  c->host_fn = fn;
  c->host_user = user;
  c->world = world;
  c->rank = rank;
  return 0;
}

int povar_comm_init(povar_ctx* c, int32_t world, int32_t rank, const uint8_t id[128]) {
  if (int rc = check_ctx(c)) return rc;
  if (world < 1 || rank < 0 || rank >= world) return fail(-1, "bad communicator arguments");
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  NCCL_TRY(ncclCommInitRank(&comm, world, u, rank));  // e.g. two ranks on one device: "Duplicate GPU detected"
  c->comm = comm;
  c->world = world;
  c->rank = rank;
  return 0;
}

}  // extern "C"
