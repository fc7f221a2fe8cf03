// host_comm.cpp -- the path's ONE collective behind the C ABI: an RCCL communicator per process (one process per
// GPU) and the all-reduce (sum, uint64) of the device-resident counter vector on the launch stream.
//
// Reference counterpart: the host-side pooling of per-worker counters, merge_power_results
// (crates/engine-orchestrator/src/merge.rs:26-70) over the counters aggregate_batch folds (aggregation.rs:61-330).
// There a worker pool hands its counters to the host, which adds them; here every rank's counters stay in HBM, RCCL
// adds them over xGMI, and the one device-to-host copy of the call brings back the pooled vector.
//
// librccl is bound lazily (dlopen at the first communicator call): a single-GPU host never maps its ~570 MB of code
// objects.  No torch, no MPI: the unique id travels over whatever channel the host already has.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <string>

#include "../../include/mcpower_hip.h"
#include "host_internal.hpp"

static_assert(sizeof(mcp_comm_id) == sizeof(ncclUniqueId), "mcp_comm_id carries an ncclUniqueId verbatim");

namespace mcp {
namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = -1, g_world = 0, g_comm_device = -1;
unsigned long long* g_warm = nullptr;  // one device word: the bring-up all-reduce of comm_init
// Every collective of the communicator is issued on ONE stream, in call order: a batch's launch stream hands over with
// an event, the all-reduce runs on g_comm_stream, a second event hands back.  Batches in flight launch on their own
// streams (host_api.hip); RCCL would accept collectives of one communicator from several streams, but a single issue
// stream makes the order of the rings the order of the calls on every rank by construction.
hipStream_t g_comm_stream = nullptr;
constexpr int COMM_EVENTS = 16;  // > batches in flight per process; an event pair is reused COMM_EVENTS / 2 calls later
hipEvent_t g_comm_ev[COMM_EVENTS] = {};
unsigned g_comm_ev_next = 0;

bool load_rccl(std::string& err) {
  if (g_rccl.handle) return true;
  // RCCL must sit on the SAME HIP runtime as this library (streams and device pointers cross the boundary), so it is
  // taken from the directory of the libamdhip64 this library is bound to, by absolute path.  A bare soname would
  // return whatever librccl.so.1 the process already holds -- under Python that is torch's bundled copy, which runs
  // on torch's own bundled HIP runtime and fails every call made with this runtime's handles.
  std::string tried;
  void* h = nullptr;
  Dl_info info;
  // MCP_RCCL_PATH names the library outright (a deployment with RCCL outside the runtime's directory); nothing else is tried
  if (const char* forced = std::getenv("MCP_RCCL_PATH")) {
    tried = std::string(forced) + " ";
    h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      const char* why = dlerror();
      err = std::string("RCCL is not loadable (tried ") + tried + "from MCP_RCCL_PATH): " + (why ? why : "not found");
      return false;
    }
  } else if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      for (const char* n : {"librccl.so.1", "librccl.so"}) {
        const std::string path = dir + n;
        tried += path + " ";
        h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) break;
      }
    }
  }
  if (!h) {
    for (const char* n : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1"}) {
      tried += std::string(n) + " ";
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
  }
  if (!h) {
    const char* why = dlerror();  // the call clears the state: read it once
    err = std::string("RCCL is not loadable (tried ") + tried + "): " + (why ? why : "not found");
    return false;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
  r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
  r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
  r.error_string = (decltype(r.error_string))dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
    dlclose(h);
    err = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
    return false;
  }
  g_rccl = r;
  return true;
}

void release_comm_stream() {  // caller holds g_mu
  for (hipEvent_t& e : g_comm_ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  if (g_comm_stream) (void)hipStreamDestroy(g_comm_stream);
  g_comm_stream = nullptr;
  if (g_warm) (void)hipFree(g_warm);
  g_warm = nullptr;
}

std::string rccl_msg(const char* what, ncclResult_t rc) {
  return std::string("RCCL error at ") + what + ": " + g_rccl.error_string(rc);
}

}  // namespace

// rank r of W owns the contiguous replicate slice [r*S/W, (r+1)*S/W): the slices partition [0, S) for every (S, W)
// and all ranks keep the SAME base_seed, so the union is bit-identical to a one-device run.
// 128-bit products: S up to 2^64 - 1 with W up to 2^31.
void shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t* offset, uint64_t* count) {
  const unsigned __int128 lo = (unsigned __int128)n_total * rank / world;
  const unsigned __int128 hi = (unsigned __int128)n_total * (rank + 1u) / world;
  *offset = (uint64_t)lo;
  *count = (uint64_t)(hi - lo);
}

bool comm_active() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_comm != nullptr;
}

int comm_device() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_comm_device;
}

// in-place sum over all ranks of `n` uint64 words resident on this rank's device, enqueued on `st`
int comm_allreduce_u64(unsigned long long* d_words, size_t n, hipStream_t st, std::string& err) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm) {
    err = "invalid spec: MCP_ALLREDUCE_COUNTERS without a communicator (call mcp_comm_init first)";
    return MCP_ERR_INVALID_SPEC;
  }
  hipEvent_t ev_in = g_comm_ev[g_comm_ev_next], ev_out = g_comm_ev[g_comm_ev_next + 1];
  g_comm_ev_next = (g_comm_ev_next + 2u) % COMM_EVENTS;
  if (hipEventRecord(ev_in, st) != hipSuccess || hipStreamWaitEvent(g_comm_stream, ev_in, 0) != hipSuccess) {
    err = "HIP error: cannot hand the counters over to the communicator's stream";
    return MCP_ERR_DEVICE;
  }
  const ncclResult_t rc = g_rccl.all_reduce(d_words, d_words, n, ncclUint64, ncclSum, g_comm, g_comm_stream);
  if (rc != ncclSuccess) {
    err = rccl_msg("ncclAllReduce", rc);
    return MCP_ERR_DEVICE;
  }
  if (hipEventRecord(ev_out, g_comm_stream) != hipSuccess || hipStreamWaitEvent(st, ev_out, 0) != hipSuccess) {
    err = "HIP error: cannot hand the pooled counters back to the launch stream";
    return MCP_ERR_DEVICE;
  }
  return MCP_OK;
}

// host-only: can this process bind RCCL at all?  (no device needed; what mcp_comm_init would fail with otherwise)
int comm_probe(std::string& err) {
  std::lock_guard<std::mutex> lk(g_mu);
  return load_rccl(err) ? MCP_OK : MCP_ERR_DEVICE;
}

int comm_unique_id(mcp_comm_id* out, int device, std::string& err) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    err = "no HIP device visible: an RCCL communicator needs its MI355X (there is no CPU fallback)";
    return MCP_ERR_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    err = "HIP error: cannot select the device for the communicator";
    return MCP_ERR_DEVICE;
  }
  if (!load_rccl(err)) return MCP_ERR_DEVICE;
  ncclUniqueId id;
  const ncclResult_t rc = g_rccl.get_unique_id(&id);
  if (rc != ncclSuccess) {
    err = rccl_msg("ncclGetUniqueId", rc);
    return MCP_ERR_DEVICE;
  }
  std::memcpy(out->bytes, id.internal, sizeof id.internal);
  return MCP_OK;
}

int comm_init(int rank, int world, const mcp_comm_id* id, int device, std::string& err) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_comm) {
    err = "invalid spec: a communicator already exists in this process (mcp_comm_destroy it first)";
    return MCP_ERR_INVALID_SPEC;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    err = "no HIP device visible: an RCCL communicator needs its MI355X (there is no CPU fallback)";
    return MCP_ERR_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    err = "HIP error: cannot select the device for the communicator";
    return MCP_ERR_DEVICE;
  }
  if (!load_rccl(err)) return MCP_ERR_DEVICE;
  ncclUniqueId nid;
  std::memcpy(nid.internal, id->bytes, sizeof nid.internal);
  ncclComm_t comm = nullptr;
  ncclResult_t rc = g_rccl.comm_init_rank(&comm, world, nid, rank);
  if (rc != ncclSuccess) {
    err = rccl_msg("ncclCommInitRank", rc);
    return MCP_ERR_DEVICE;
  }
  // bring the rings up now (the first collective of a communicator builds its channels over xGMI), so that the first
  // batch of the caller does not pay for it; also proves the world agrees before any batch is enqueued
  unsigned long long one = 1ull, back = 0ull;
  bool ok = hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok && i < COMM_EVENTS; ++i) ok = hipEventCreateWithFlags(&g_comm_ev[i], hipEventDisableTiming) == hipSuccess;
  g_comm_ev_next = 0;
  ok = ok && hipMalloc((void**)&g_warm, sizeof one) == hipSuccess &&
       hipMemcpy(g_warm, &one, sizeof one, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    rc = g_rccl.all_reduce(g_warm, g_warm, 1, ncclUint64, ncclSum, comm, g_comm_stream);
    ok = rc == ncclSuccess && hipStreamSynchronize(g_comm_stream) == hipSuccess &&
         hipMemcpy(&back, g_warm, sizeof back, hipMemcpyDeviceToHost) == hipSuccess;
  }
  if (!ok || back != (unsigned long long)world) {
    err = !ok ? std::string("the bring-up all-reduce of the communicator failed")
              : std::string("the bring-up all-reduce returned ") + std::to_string(back) + " for a world of " + std::to_string(world);
    (void)g_rccl.comm_destroy(comm);
    release_comm_stream();
    return MCP_ERR_DEVICE;
  }
  g_comm = comm;
  g_rank = rank;
  g_world = world;
  g_comm_device = device;
  return MCP_OK;
}

int comm_destroy(std::string& err) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm) return MCP_OK;
  (void)hipSetDevice(g_comm_device);
  (void)hipDeviceSynchronize();  // no collective of this communicator may still be queued
  const ncclResult_t rc = g_rccl.comm_destroy(g_comm);
  g_comm = nullptr;
  g_rank = -1;
  g_world = 0;
  g_comm_device = -1;
  release_comm_stream();
  if (rc != ncclSuccess) {
    err = rccl_msg("ncclCommDestroy", rc);
    return MCP_ERR_DEVICE;
  }
  return MCP_OK;
}

int comm_rank() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_rank;
}

int comm_world() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_world;
}

}  // namespace mcp
