// mnk_exchange.hip -- the one exchange step of the multi-GPU rollout path: an RCCL all-gather of a rank's
// rollout message (packed records, or chunk-start state + action log) over xGMI, on the caller's HIP stream.
//
// The reference has no distributed code; what is exchanged is the content of its RolloutBuffer
// (alg/rollout_buffer.py:14-44) in this build's packed forms (include/mnk_hip.h).  Envs shard by contiguous
// blocks of global env ids and nothing else on the path communicates.
//
// RCCL is not linked: the process that calls this already has one RCCL loaded (PyTorch-ROCm ships its own
// librccl.so.1, which torch.distributed's "nccl" backend uses), and a second copy in the same address space would
// duplicate its global state.  The entry points are therefore resolved at first use from the librccl.so.1 that is
// already mapped (RTLD_NOLOAD), falling back to the ROCm installation's when none is.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mnk_host.h"

namespace {

struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclGetVersion) get_version = nullptr;
  // the direct form of the exchange: one send and one receive per peer, grouped
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;
  decltype(&ncclCommUserRank) comm_user_rank = nullptr;
  bool ok = false;
  bool p2p_ok = false;
};

thread_local char g_comm_err[256] = "";

const Rccl& rccl() {
  static Rccl r = [] {
    Rccl t;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return t;
    t.get_unique_id = (decltype(t.get_unique_id))dlsym(h, "ncclGetUniqueId");
    t.comm_init_rank = (decltype(t.comm_init_rank))dlsym(h, "ncclCommInitRank");
    t.comm_destroy = (decltype(t.comm_destroy))dlsym(h, "ncclCommDestroy");
    t.all_gather = (decltype(t.all_gather))dlsym(h, "ncclAllGather");
    t.error_string = (decltype(t.error_string))dlsym(h, "ncclGetErrorString");
    t.get_version = (decltype(t.get_version))dlsym(h, "ncclGetVersion");
    t.group_start = (decltype(t.group_start))dlsym(h, "ncclGroupStart");
    t.group_end = (decltype(t.group_end))dlsym(h, "ncclGroupEnd");
    t.send = (decltype(t.send))dlsym(h, "ncclSend");
    t.recv = (decltype(t.recv))dlsym(h, "ncclRecv");
    t.comm_count = (decltype(t.comm_count))dlsym(h, "ncclCommCount");
    t.comm_user_rank = (decltype(t.comm_user_rank))dlsym(h, "ncclCommUserRank");
    t.ok = t.get_unique_id && t.comm_init_rank && t.comm_destroy && t.all_gather && t.error_string;
    t.p2p_ok = t.ok && t.group_start && t.group_end && t.send && t.recv && t.comm_count && t.comm_user_rank;
    return t;
  }();
  return r;
}

int comm_status(const char* what, ncclResult_t rc) {
  if (rc == ncclSuccess) return MNK_OK;
  snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, rccl().error_string(rc));
  return MNK_ECOMM;
}

int need_rccl() {
  if (rccl().ok) return MNK_OK;
  snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so.1 could not be resolved: %s", dlerror());
  return MNK_ECOMM;
}

}  // namespace

extern "C" {

const char* mnk_comm_last_error(void) { return g_comm_err; }

int mnk_comm_version(void) {
  int v = 0;
  if (need_rccl() != MNK_OK || !rccl().get_version || rccl().get_version(&v) != ncclSuccess) return 0;
  return v;
}

int mnk_comm_unique_id(void* id_out_host) {
  if (!id_out_host) return MNK_EINVAL;
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  if (int rc = comm_status("ncclGetUniqueId", rccl().get_unique_id(&id))) return rc;
  memcpy(id_out_host, &id, MNK_COMM_ID_BYTES);
  return MNK_OK;
}

int mnk_comm_init(void** comm_out, const void* id_host, int nranks, int rank) {
  if (!comm_out || !id_host || nranks < 1 || rank < 0 || rank >= nranks) return MNK_EINVAL;
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, id_host, MNK_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  if (int rc = comm_status("ncclCommInitRank", rccl().comm_init_rank(&comm, nranks, id, rank))) return rc;
  *comm_out = comm;
  return MNK_OK;
}

int mnk_comm_destroy(void* comm) {
  if (!comm) return MNK_OK;
  if (int rc = need_rccl()) return rc;
  return comm_status("ncclCommDestroy", rccl().comm_destroy((ncclComm_t)comm));
}

int mnk_allgather_records(void* comm, const void* send, void* recv, int64_t bytes, void* stream) {
  if (!comm || !send || !recv || bytes < 0) return MNK_EINVAL;
  if (bytes == 0) return MNK_OK;
  if (int rc = need_rccl()) return rc;
  // 8-byte elements when the message allows it (all message parts are u64 / padded to u64): fewer, wider elements
  const bool wide = (bytes % 8 == 0) && (((uintptr_t)send | (uintptr_t)recv) % 8 == 0);
  return comm_status("ncclAllGather",
                     rccl().all_gather(send, recv, wide ? (size_t)(bytes / 8) : (size_t)bytes,
                                       wide ? ncclUint64 : ncclUint8, (ncclComm_t)comm, (hipStream_t)stream));
}

int mnk_allgather_records_direct(void* comm, const void* send, void* recv, int64_t bytes, void* stream) {
  if (!comm || !send || !recv || bytes < 0) return MNK_EINVAL;
  if (bytes == 0) return MNK_OK;
  if (int rc = need_rccl()) return rc;
  const Rccl& r = rccl();
  if (!r.p2p_ok) {
    snprintf(g_comm_err, sizeof(g_comm_err), "ncclSend / ncclRecv / ncclGroup* could not be resolved from librccl");
    return MNK_ECOMM;
  }
  int nranks = 0, rank = 0;
  if (int rc = comm_status("ncclCommCount", r.comm_count((ncclComm_t)comm, &nranks))) return rc;
  if (int rc = comm_status("ncclCommUserRank", r.comm_user_rank((ncclComm_t)comm, &rank))) return rc;
  const bool wide = (bytes % 8 == 0) && (((uintptr_t)send | (uintptr_t)recv) % 8 == 0);
  const size_t count = wide ? (size_t)(bytes / 8) : (size_t)bytes;
  const ncclDataType_t type = wide ? ncclUint64 : ncclUint8;
  // every rank posts its sends and receives in the same rotated order (peer = rank + i): at step i of the rotation every
  // link of the mesh carries exactly one message in each direction.  The rank's own slot goes through the same calls
  // (RCCL turns a send to oneself into a local copy), so a one-rank communicator exercises the whole path.
  if (int rc = comm_status("ncclGroupStart", r.group_start())) return rc;
  ncclResult_t first = ncclSuccess;
  for (int i = 0; i < nranks; ++i) {
    const int to = (rank + i) % nranks, from = (rank - i + nranks) % nranks;
    ncclResult_t a = r.send(send, count, type, to, (ncclComm_t)comm, (hipStream_t)stream);
    ncclResult_t b = r.recv((char*)recv + (size_t)from * (size_t)bytes, count, type, from, (ncclComm_t)comm,
                            (hipStream_t)stream);
    if (first == ncclSuccess) first = a != ncclSuccess ? a : b;
  }
  ncclResult_t end = r.group_end();  // always closed, also after a failed post
  if (first != ncclSuccess) return comm_status("ncclSend / ncclRecv", first);
  return comm_status("ncclGroupEnd", end);
}

}  // extern "C"
