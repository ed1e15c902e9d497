// tbc_comm.hip -- the sharded sweep's ONE exchange behind the C-ABI (include/tbcheck.h, "one history over several GPUs": tbc_comm_*,
// tbc_batch_sweep_allgather).
//
// Until round 5 the all-gather of the ranks' relation tables lived in Python (jepsen-tigerbeetle_amd/shard.py over torch.distributed):
// correct, tested over gloo, and unreachable from the host the reference actually is -- a Clojure process (project.clj:6-8) that binds
// this library through JNA.  Here a rank of ANY host language does
//
//     tbc_comm_unique_id(id)                       rank 0; the 128 bytes travel to the other ranks however the host likes
//     tbc_comm_init(rank, world, id, device, &c)   RCCL communicator (librccl.so is dlopen'ed here, at the first call: a single-GPU
//                                                  caller never loads it, and the library still loads where RCCL is not installed)
//     tbc_batch_sweep_allgather(batch, c, results) this rank's share of the sweep's wavefronts, ONE ncclAllGather of the relation
//                                                  tables straight out of HBM over xGMI, OR-merge on the device, composition
//
// -- or, with tbc_comm_init_host, the same through a caller-supplied all-gather over HOST memory (its own fabric, MPI, a test's gloo):
// the table comes down, the callback gathers, the merged table goes back through tbc_batch_sweep_finish.  That transport is what the
// two-process test drives (tests/test_comm_gpu.py) -- one GPU box is all this pool has; the RCCL path is exercised there at world 1.
// xGMI is point-to-point (7 links x ~153 GB/s a GPU); an all-gather of ~700 KB a rank is latency, not bandwidth: one collective, once.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "tbc_batch.h"

using namespace tbc;

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // (RTLD_GLOBAL is not wanted: a process that has torch loaded has torch's own librccl mapped already, and dlopen by soname hands
    // that very object back -- one RCCL per process either way)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy && r.GetErrorString;
  });
  return r;
}

#define NCCL_TRY(expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != ncclSuccess) { set_error("%s failed: %s", #expr, rccl().GetErrorString(r_)); return TBC_ERR_HIP; } \
  } while (0)

}  // namespace

struct tbc_comm {
  uint32_t rank = 0, world = 1;
  int device = 0;
  ncclComm_t nccl = nullptr;                 // RCCL transport ...
  tbc_allgather_fn host_fn = nullptr;        // ... or the caller's, over host memory
  void* host_user = nullptr;
  DevBuf<uint8_t> gathered;                  // RCCL: world tables back to back, in HBM
  std::vector<uint8_t> host_send, host_recv;
};

extern "C" {

tbc_status tbc_comm_unique_id(void* id) {
  if (!id) { set_error("tbc_comm_unique_id: null argument"); return TBC_ERR_INVALID_ARG; }
  if (!rccl().ok) { set_error("librccl.so could not be loaded (%s): no RCCL transport on this machine", dlerror() ? dlerror() : "symbols missing"); return TBC_ERR_UNSUPPORTED; }
  static_assert(TBC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "tbc_comm_unique_id hands out an ncclUniqueId");
  ncclUniqueId u;
  NCCL_TRY(rccl().GetUniqueId(&u));
  std::memcpy(id, u.internal, TBC_COMM_ID_BYTES);
  return TBC_OK;
}

tbc_status tbc_comm_init(uint32_t rank, uint32_t world, const void* id, uint32_t device, tbc_comm** out) {
  if (!id || !out || world == 0 || rank >= world) { set_error("tbc_comm_init: bad rank / world / null argument"); return TBC_ERR_INVALID_ARG; }
  if (!rccl().ok) { set_error("librccl.so could not be loaded: no RCCL transport on this machine"); return TBC_ERR_UNSUPPORTED; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || (int)device >= ndev || !device_is_gfx950((int)device)) { set_error("device %u is not a gfx950 (MI355X) device", device); return TBC_ERR_NO_DEVICE; }
  HIP_TRY(hipSetDevice((int)device));
  tbc_comm* c = new (std::nothrow) tbc_comm();
  if (!c) return TBC_ERR_OOM;
  c->rank = rank; c->world = world; c->device = (int)device;
  ncclUniqueId u;
  std::memcpy(u.internal, id, TBC_COMM_ID_BYTES);
  const ncclResult_t r = rccl().CommInitRank(&c->nccl, (int)world, u, (int)rank);
  if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(r)); delete c; return TBC_ERR_HIP; }
  *out = c;
  return TBC_OK;
}

tbc_status tbc_comm_init_host(uint32_t rank, uint32_t world, tbc_allgather_fn fn, void* user, tbc_comm** out) {
  if (!fn || !out || world == 0 || rank >= world) { set_error("tbc_comm_init_host: bad rank / world / null argument"); return TBC_ERR_INVALID_ARG; }
  tbc_comm* c = new (std::nothrow) tbc_comm();
  if (!c) return TBC_ERR_OOM;
  c->rank = rank; c->world = world; c->host_fn = fn; c->host_user = user;
  *out = c;
  return TBC_OK;
}

uint32_t tbc_comm_rank(const tbc_comm* c) { return c ? c->rank : 0; }
uint32_t tbc_comm_world(const tbc_comm* c) { return c ? c->world : 0; }

void tbc_comm_destroy(tbc_comm* c) {
  if (!c) return;
  if (c->nccl) { (void)hipSetDevice(c->device); (void)rccl().CommDestroy(c->nccl); }
  c->gathered.release();
  delete c;
}

// this rank's share of ONE sharded check: its wavefronts swept, the tables exchanged once, every rank composes.  Every rank of the
// communicator must call it with a batch created from the SAME histories and options (the inputs are replicated: a history is a few hundred KB).
tbc_status tbc_batch_sweep_allgather(tbc_batch* b, tbc_comm* c, tbc_result* results) {
  if (!b || !c) { set_error("tbc_batch_sweep_allgather: null argument"); return TBC_ERR_INVALID_ARG; }
  try {
    tbc_status s = tbc_batch_set_shard(b, c->rank, c->world);
    if (s != TBC_OK) return s;
    if ((s = tbc_batch_sweep_partial(b)) != TBC_OK) return s;
    void* table = nullptr;
    uint64_t bytes = 0;
    if ((s = tbc_batch_sweep_table(b, &table, &bytes)) != TBC_OK) return s;
    if (c->nccl) {
      if (c->device != b->device) { set_error("tbc_batch_sweep_allgather: the communicator lives on device %d, the batch on %d", c->device, b->device); return TBC_ERR_INVALID_ARG; }
      HIP_TRY(hipSetDevice(b->device));
      if (c->gathered.n < bytes * c->world) { c->gathered.release(); if ((s = c->gathered.alloc((size_t)(bytes * c->world))) != TBC_OK) return s; }
      // (tbc_batch_sweep_partial has synchronised the batch's stream: the table is complete; the collective runs on that stream and
      // tbc_batch_sweep_merge's OR kernel follows it there)
      NCCL_TRY(rccl().AllGather(table, c->gathered.p, (size_t)bytes, ncclUint8, c->nccl, b->stream));
      return tbc_batch_sweep_merge(b, c->gathered.p, bytes * c->world, c->world, results);
    }
    // the caller's transport, over host memory
    c->host_send.resize((size_t)bytes);
    c->host_recv.resize((size_t)(bytes * c->world));
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(c->host_send.data(), table, (size_t)bytes, hipMemcpyDeviceToHost));
    const int rc = c->host_fn(c->host_user, c->host_send.data(), c->host_recv.data(), bytes);
    if (rc != 0) { set_error("tbc_batch_sweep_allgather: the caller's all-gather returned %d", rc); return TBC_ERR_HIP; }
    // every record is written by exactly one rank and all zero on the others: the merged table is the bitwise OR
    uint64_t* acc = reinterpret_cast<uint64_t*>(c->host_send.data());
    std::memset(acc, 0, (size_t)bytes);
    for (uint32_t r = 0; r < c->world; r++) {
      const uint64_t* src = reinterpret_cast<const uint64_t*>(c->host_recv.data() + (size_t)r * bytes);
      for (uint64_t i = 0; i < bytes / 8; i++) acc[i] |= src[i];
    }
    return tbc_batch_sweep_finish(b, c->host_send.data(), bytes, results);
  } catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

}  // extern "C"
