// host_api.hip -- the C ABI (include/mcpower_hip.h) over the gfx950 kernels.
//
// Host orchestration of run_batch (reference crates/engine-core/src/batch.rs:199-776):
// validate -> crit tables -> result tensors -> dispatch replicates -> fold.  The
// reference dispatches replicates over a rayon pool (:637-742); here the replicate
// range is cut into device launches ("chunks") on one HIP stream, the progress /
// cancel callback runs between chunks (:2149-2177), and the additive counters the
// orchestrator would fold (aggregation.rs:84-284) are accumulated on the device.
// There is NO CPU fallback: anything the kernels do not cover returns
// MCP_ERR_UNSUPPORTED, and a missing device returns MCP_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcpower_hip.h"
#include "host_internal.hpp"
#include "mcp_params.hpp"

namespace mcp {
hipError_t launch_ols_fused(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, int flavor, uint32_t n_grid, hipStream_t st);
size_t ols_rec_fields(int P);
size_t ols_wide_rec_fields(int P);  // ols_wide_kernels.hip: the whole triangle of [1 | X | y]'[1 | X | y]
hipError_t launch_design_capture(int P, const SimParams* d_prm, uint32_t n_sims, bool scen, hipStream_t st);
hipError_t launch_ols_reduced(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, uint32_t between_floats, hipStream_t st);
hipError_t launch_ols_generate(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, hipStream_t st);
hipError_t launch_scenario_design(const SimParams* d_prm, uint32_t n_sims, hipStream_t st, uint32_t* d_hist = nullptr);
hipError_t launch_ols_fit(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, hipStream_t st);
hipError_t launch_ols_fit_provided(int P, const float* design, const float* outcome, uint32_t n_fits, uint32_t n,
                                   double* beta, double* se, double* tsq, double* rss, double* sst, uint8_t* conv,
                                   hipStream_t st);
hipError_t launch_icdf_probe(uint32_t first_word, uint32_t stride, uint32_t n, float* out, hipStream_t st);
hipError_t launch_glm_sim(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, bool factors, bool reduced,
                          bool scen, bool logit, hipStream_t st);
hipError_t launch_fold_flags(const SimParams* d_prm, uint32_t n_sims, hipStream_t st);
size_t glm_lds_bytes(int P, uint32_t max_n);
hipError_t launch_lmm_sim(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max, bool factors,
                          bool reduced, bool scen, hipStream_t st);
size_t lmm_lds_bytes(int P, uint32_t max_n, uint32_t ncl_max);
size_t lmm_rec_fields(int P, uint32_t ncl_max);
hipError_t launch_glmm(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max, bool scen, hipStream_t st);
size_t glmm_lds_bytes(int P, uint32_t max_n, uint32_t ncl_max);
hipError_t launch_glmm_slopes(int P, int Q, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max, bool scen,
                              hipStream_t st);
size_t glmm_slopes_lds_bytes(int P, int Q, uint32_t max_n, uint32_t ncl_max);
hipError_t launch_glmm_general(int P, const SimParams* d_prm, uint32_t n_sims, size_t lds, bool scen, hipStream_t st);
size_t glmm_general_lds_bytes(const SimParams& prm);
hipError_t launch_lmm_general(int P, int q0, const SimParams* d_prm, uint32_t n_sims, size_t lds, hipStream_t st);
hipError_t launch_lmm_general_scen(int P, int q0, const SimParams* d_prm, uint32_t n_sims, size_t lds, hipStream_t st);
size_t lmg_lds_bytes(const SimParams& prm);
hipError_t launch_reduce_counters(const unsigned long long* copies, uint32_t copy_stride, uint32_t n,
                                  unsigned long long* out, hipStream_t st);
int comm_allreduce_u64(unsigned long long* d_words, size_t n, hipStream_t st, std::string& err);  // host_comm.cpp
}  // namespace mcp

namespace {

thread_local std::string g_err;
// the device of this process (one process per GPU); set once by mcp_set_device before the first batch, read by every
// host thread -- arenas, tickets and streams are per thread and rebuilt when it changes
std::atomic<int> g_device{0};

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess)                                                                                      \
      return fail(MCP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);          \
  } while (0)

// grow-only device buffers reused across calls of one host thread (no hipMalloc on the steady-state path)
struct Arena {
  struct Buf {
    void* p = nullptr;
    size_t cap = 0;
  };
  Buf slots[32];
  Buf pinned[2];  // host staging: [0] the upload blob, [1] the counters coming back
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;
  int device = -1;

  hipError_t get_pinned(int slot, size_t bytes, void** out) {
    Buf& b = pinned[slot];
    if (bytes > b.cap) {
      if (b.p) (void)hipHostFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      const size_t want = bytes + (bytes >> 2) + 4096;
      hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      b.cap = want;
    }
    *out = b.p;
    return hipSuccess;
  }

  hipError_t get(int slot, size_t bytes, void** out) {
    Buf& b = slots[slot];
    if (bytes > b.cap) {
      if (b.p) (void)hipFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      const size_t want = bytes + (bytes >> 3) + 256;
      hipError_t e = hipMalloc(&b.p, want);
      if (e != hipSuccess) return e;
      b.cap = want;
    }
    *out = b.p;
    return hipSuccess;
  }
  // give everything back to the device the arena was built on (device switch; the slots are grow-only otherwise)
  void release() {
    if (device < 0) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
    for (Buf& b : slots)
      if (b.p) (void)hipFree(b.p);
    for (Buf& b : pinned)
      if (b.p) (void)hipHostFree(b.p);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    *this = Arena();
    if (cur >= 0) (void)hipSetDevice(cur);
  }
  hipError_t event(size_t i, hipEvent_t* ev) {
    while (events.size() <= i) {
      hipEvent_t e;
      hipError_t rc = hipEventCreate(&e);
      if (rc != hipSuccess) return rc;
      events.push_back(e);
    }
    *ev = events[i];
    return hipSuccess;
  }
};
thread_local Arena g_arena;
// batches in flight (mcp_run_batch_submit / _collect): each ticket owns an arena AND a stream, so the kernels of
// successive batches may overlap at their edges (tail of one, head of the next); a caller-provided stream
// (mcp_run_options.hip_stream) keeps them strictly in order instead
constexpr int MAX_IN_FLIGHT = 4;
struct Ticket {
  bool busy = false;
  Arena arena;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_upload = nullptr;
  uint64_t* h_ctr = nullptr;
  uint32_t n_ctr = 0, launched = 0;
  size_t ev_per_chunk = 0;
  bool glm = false, reduced = false, staged = false;
  mcp_batch_result shell;
};
thread_local Ticket g_tickets[MAX_IN_FLIGHT];
thread_local hipStream_t g_upload_stream = nullptr;  // a batch in flight uploads its tables beside the running kernel
thread_local hipEvent_t g_epoch = nullptr;           // recorded once per host thread: origin of kernel_begin_ms / kernel_end_ms

enum Slot { S_PRM = 0, S_CRIT, S_LEVELS, S_COUNTERS, S_UNC, S_COR, S_CONV, S_OVERALL, S_STAT, S_TILE, S_MISC0, S_MISC1, S_BOUNDARY, S_JOINT, S_SCEN_L, S_SCEN_META, S_SCEN_BETA, S_T3, S_PH_UNC, S_PH_COR, S_FX, S_CRIT_RED, S_RED_ROW, S_SCEN_TAU, S_UPLOAD_NORMAL, S_UPLOAD_DATA, S_LMM_REC, S_LMM_FLAGS, S_PINNED, S_TAU_HAT, S_SCEN_HIST, S_SCEN_PERM };

int ensure_device(hipStream_t* st, const mcp_run_options* opts, Arena* extra = nullptr) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(MCP_ERR_DEVICE, "no HIP device visible: the MI355X path cannot run (there is no CPU fallback)");
  const int dev = g_device.load();
  if (dev >= n) return fail(MCP_ERR_DEVICE, "selected device index out of range");
  HIP_TRY(hipSetDevice(dev));
  if (g_arena.device != dev) {
    g_arena.release();
    g_arena.device = dev;
    // the timing origin and the side stream belong to the device they were created on
    if (g_epoch) (void)hipEventDestroy(g_epoch);
    g_epoch = nullptr;
    if (g_upload_stream) (void)hipStreamDestroy(g_upload_stream);
    g_upload_stream = nullptr;
  }
  if (extra && extra->device != dev) {
    extra->release();
    extra->device = dev;
  }
  if (opts && opts->hip_stream) {
    *st = (hipStream_t)opts->hip_stream;
  } else if (extra) {
    // a batch in flight launches on its ticket's own stream: the last, partly filled round of resident waves of batch i
    // (a 10^6-replicate launch is 3.05 rounds of 1 280 workgroups) then shares the chip with the head of batch i + 1
    // instead of running alone on a few CUs
    if (!extra->stream) HIP_TRY(hipStreamCreateWithFlags(&extra->stream, hipStreamNonBlocking));
    *st = extra->stream;
  } else {
    if (!g_arena.stream) HIP_TRY(hipStreamCreateWithFlags(&g_arena.stream, hipStreamNonBlocking));
    *st = g_arena.stream;
  }
  return MCP_OK;
}

template <class T>
T* host_alloc(size_t n) {
  return (T*)std::malloc((n ? n : 1) * sizeof(T));
}

}  // namespace

extern "C" {

uint32_t mcp_abi_version(void) { return MCP_ABI_VERSION; }
const char* mcp_last_error(void) { return g_err.c_str(); }

int mcp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mcp_set_device(int device_index) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MCP_ERR_DEVICE, "no HIP device visible");
  if (device_index < 0 || device_index >= n) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: device index out of range");
  g_device = device_index;
  return MCP_OK;
}

int mcp_comm_unique_id(mcp_comm_id* id_out) {
  if (!id_out) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: null communicator id pointer");
  std::string err;
  const int rc = mcp::comm_unique_id(id_out, g_device, err);
  return rc ? fail(rc, err) : MCP_OK;
}

int mcp_comm_init(int rank, int world, const mcp_comm_id* id) {
  if (!id) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: null communicator id pointer");
  if (world < 1 || rank < 0 || rank >= world)
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: communicator rank must lie in [0, world) with world >= 1");
  std::string err;
  const int rc = mcp::comm_init(rank, world, id, g_device, err);
  return rc ? fail(rc, err) : MCP_OK;
}

int mcp_comm_destroy(void) {
  std::string err;
  const int rc = mcp::comm_destroy(err);
  return rc ? fail(rc, err) : MCP_OK;
}

int mcp_comm_probe(void) {
  std::string err;
  const int rc = mcp::comm_probe(err);
  return rc ? fail(rc, err) : MCP_OK;
}

int mcp_comm_rank(void) { return mcp::comm_rank(); }
int mcp_comm_world(void) { return mcp::comm_world(); }

int mcp_shard_range(uint64_t n_sims_total, int rank, int world, uint64_t* offset_out, uint64_t* count_out) {
  if (!offset_out || !count_out) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: null shard output pointer");
  if (world < 1 || rank < 0 || rank >= world)
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: shard rank must lie in [0, world) with world >= 1");
  mcp::shard_range(n_sims_total, (uint32_t)rank, (uint32_t)world, offset_out, count_out);
  return MCP_OK;
}

double mcp_t_ppf(double p, double df) { return mcp::t_ppf(p, df); }
double mcp_norm_ppf(double p) { return mcp::norm_ppf(p); }
double mcp_chi2_ppf(double p, double k) { return mcp::chi2_ppf(p, k); }
double mcp_f_ppf(double p, double dfn, double dfd) { return mcp::f_ppf(p, dfn, dfd); }

int mcp_build_crit_table(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, double* crit_out) {
  std::string err;
  int rc = mcp::validate_spec(spec, ns, n_ns, err);
  if (rc) return fail(rc, err);
  const uint32_t P = 1 + spec->n_non_factor + spec->n_factor_dummies + spec->n_interactions;
  const uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs;
  std::vector<double> rows, kk;
  if (spec->correction_method == MCP_CORR_TUKEY_HSD) mcp::tukey_k_per_target(spec, kk);
  mcp::TukeyMemo memo;
  rc = mcp::build_crit_rows(spec->estimator, spec->correction_method, spec->alpha, spec->posthoc_alpha, P, nt, ns, n_ns,
                            kk.empty() ? nullptr : kk.data(), &memo, rows, err);
  if (rc) return fail(rc, err);
  std::memcpy(crit_out, rows.data(), rows.size() * sizeof(double));
  return MCP_OK;
}

void mcp_free_result(mcp_batch_result* r) {
  if (!r) return;
  std::free(r->uncorrected); std::free(r->corrected); std::free(r->posthoc_unc); std::free(r->posthoc_cor);
  std::free(r->converged); std::free(r->boundary_hit); std::free(r->pinned_components); std::free(r->joint_unc);
  std::free(r->joint_cor); std::free(r->overall); std::free(r->factor_excluded); std::free(r->tau_squared_hat);
  std::free(r->stat_capture); std::free(r->counters);
  std::memset(r, 0, sizeof *r);
}

}  // extern "C"

namespace {

// fills the timing fields of a finished batch from the events recorded around its launches
int finish_timing(Arena& ar, hipEvent_t ev_begin, hipEvent_t ev_end, uint32_t launched, size_t ev_per_chunk, bool glm,
                  bool reduced, bool staged, mcp_batch_result* out) {
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
  out->device_ms = ms;
  double kernel_ms = 0.0, aux_ms = 0.0;
  for (uint32_t c = 0; c < launched; ++c) {
    hipEvent_t e0 = ar.events[2 + c * ev_per_chunk], e1 = ar.events[3 + c * ev_per_chunk];
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (glm || reduced) {
      kernel_ms += ms;  // simulate kernel
      HIP_TRY(hipEventElapsedTime(&ms, e1, ar.events[4 + c * ev_per_chunk]));
      aux_ms += ms;  // counter fold
    } else if (!staged) {
      kernel_ms += ms;
    } else {
      aux_ms += ms;  // generator
      HIP_TRY(hipEventElapsedTime(&ms, e1, ar.events[4 + c * ev_per_chunk]));
      kernel_ms += ms;  // fit
    }
  }
  out->kernel_ms = kernel_ms;
  out->aux_kernel_ms = aux_ms;
  out->n_kernel_launches = launched;
  if (launched && g_epoch) {  // dominant kernel: the simulate / fused kernel, or the fit kernel of the staged pipeline
    const size_t last = (size_t)(launched - 1) * ev_per_chunk;
    hipEvent_t kb = staged ? ar.events[3] : ar.events[2];
    hipEvent_t ke = staged ? ar.events[4 + last] : ar.events[3 + last];
    float b_ms = 0.f, e_ms = 0.f;  // an origin from another device or stream state: no span rather than a failed batch
    if (hipEventElapsedTime(&b_ms, g_epoch, kb) == hipSuccess && hipEventElapsedTime(&e_ms, g_epoch, ke) == hipSuccess) {
      out->kernel_begin_ms = b_ms;
      out->kernel_end_ms = e_ms;
    } else {
      (void)hipGetLastError();
    }
  }
  return MCP_OK;
}

// `defer` != nullptr: counters-only batch left in flight on the ticket's arena (no host wait here)
int run_batch_impl(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t n_sims, uint64_t base_seed,
                   const mcp_run_options* opts, mcp_progress_fn progress, void* progress_user, mcp_batch_result* out,
                   Ticket* defer) {
  if (!out) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: null result pointer");
  std::memset(out, 0, sizeof *out);
  std::string err;
  int rc = mcp::validate_spec(spec, ns, n_ns, err);
  if (rc) return fail(rc, err);
  mcp::SimParams prm;
  mcp::Lowered low;
  rc = mcp::lower_spec(spec, ns, n_ns, prm, low, err);
  if (rc) return fail(rc, err);

  const uint32_t flags = opts ? opts->flags : (uint32_t)MCP_WANT_PER_SIM;
  const bool per_sim = flags & MCP_WANT_PER_SIM;
  const bool want_stat = flags & MCP_WANT_STAT_CAPTURE;
  // pool the counters over the ranks of this process's communicator (merge.rs:26-70): checked before anything is
  // enqueued, so a rank without a communicator fails here and not inside the collective
  const bool pool = flags & MCP_ALLREDUCE_COUNTERS;
  if ((flags & MCP_SEED_PER_REPLICATE) && !(spec->estimator == MCP_EST_GLM && spec->cluster.present))
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: MCP_SEED_PER_REPLICATE outside the clustered GLM (it is the validation aid of oracle2_power.R's per-draw leg)");
  if ((flags & MCP_SEED_PER_REPLICATE) && prm.scen_active)  // the scenario / heterogeneity streams key on sim_id as well
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: MCP_SEED_PER_REPLICATE with scenario perturbations");
  prm.seed_per_replicate = (flags & MCP_SEED_PER_REPLICATE) ? 1u : 0u;
  if (pool && !mcp::comm_active())
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: MCP_ALLREDUCE_COUNTERS without a communicator (call mcp_comm_init first)");
  if (pool && mcp::comm_device() != g_device)
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: the communicator was created on another device than the one selected now");
  const bool mle = spec->estimator == MCP_EST_MLE;
  const bool glm = spec->estimator == MCP_EST_GLM || mle;  // wave-per-replicate kernels + flag fold
  const bool reduced = prm.red_active && !glm;             // reduced-design OLS kernel, flags folded afterwards
  // RED variants of the wave-per-replicate kernels; their general-path (scenario) variants build on them
  // (probit / Poisson run on the superset kernel as well: only the logit kernels are family-specialised)
  const bool glm_reduced = (prm.red_active || prm.scen_active || (!mle && prm.glm_family != 0u)) && glm && !prm.glmm_active;
  if (reduced && (flags & MCP_PIPELINE_STAGED))
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: the staged pipeline with sparse-factor exclusion / test_formula / post-hoc rounds or more than 8 design columns / tests");
  const bool staged = (flags & MCP_PIPELINE_STAGED) && !glm && low.P <= (uint32_t)mcp::MAX_P_WAVE;  // the tile kernels exist for P <= 8
  const uint32_t npc = prm.n_posthoc;
  const uint64_t sim_offset = opts ? opts->sim_offset : 0;
  const uint32_t nt = low.n_tests, P = low.P;
  const size_t cells = (size_t)n_sims * n_ns;

  mcp_counter_layout layout;
  mcp::counter_layout(spec, &layout);
  out->layout = layout;
  out->n_sims = n_sims; out->n_sample_sizes = n_ns; out->n_targets = nt; out->n_factors = spec->n_factors;
  out->n_posthoc_contrasts = npc;
  out->n_variance_components = (spec->estimator == MCP_EST_MLE || (spec->estimator == MCP_EST_GLM && spec->cluster.present)) ? 1u : 0u;
  out->counters = host_alloc<uint64_t>((size_t)n_ns * layout.stride);
  std::memset(out->counters, 0, (size_t)n_ns * layout.stride * sizeof(uint64_t));
  if (n_sims == 0 && !pool) return MCP_OK;  // batch.rs:390-421: empty tensors, shape carried
  if (n_sims == 0) {
    // an empty shard (fewer replicates than ranks) still owes the other ranks its zero contribution: no kernel, a
    // zeroed vector joins the collective and the pooled counters come back like on every other rank
    hipStream_t st0 = nullptr;
    rc = ensure_device(&st0, opts, defer ? &defer->arena : nullptr);
    if (rc) { mcp_free_result(out); return rc; }
    Arena& ar0 = defer ? defer->arena : g_arena;
    const uint32_t n_ctr0 = n_ns * layout.stride;
    unsigned long long* d_vec = nullptr;
    uint64_t* h_vec = nullptr;
    hipEvent_t eb = nullptr, ee = nullptr;
    std::string cerr;
    int crc = MCP_OK;
    hipError_t he = ar0.get(S_COUNTERS, ((size_t)n_ctr0 + 1u) * sizeof(uint64_t), (void**)&d_vec);
    if (he == hipSuccess) he = ar0.get_pinned(1, ((size_t)n_ctr0 + 1u) * sizeof(uint64_t), (void**)&h_vec);
    if (he == hipSuccess) he = ar0.event(0, &eb);
    if (he == hipSuccess) he = ar0.event(1, &ee);
    if (he == hipSuccess) he = hipEventRecord(eb, st0);
    if (he == hipSuccess) he = hipMemsetAsync(d_vec, 0, ((size_t)n_ctr0 + 1u) * sizeof(uint64_t), st0);
    // from here on the peers are waiting: whatever fails locally, the collective is still joined
    if (d_vec) crc = mcp::comm_allreduce_u64(d_vec, (size_t)n_ctr0 + 1u, st0, cerr);
    if (he == hipSuccess && !crc) he = hipMemcpyAsync(h_vec, d_vec, ((size_t)n_ctr0 + 1u) * sizeof(uint64_t), hipMemcpyDeviceToHost, st0);
    if (he == hipSuccess && !crc) he = hipEventRecord(ee, st0);
    if (he != hipSuccess || crc) {
      (void)hipStreamSynchronize(st0);
      mcp_free_result(out);
      return crc ? fail(crc, cerr) : fail(MCP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(he) + " on an empty shard");
    }
    if (defer) {
      defer->ev_begin = eb; defer->ev_end = ee; defer->h_ctr = h_vec; defer->n_ctr = n_ctr0;
      defer->launched = 0; defer->ev_per_chunk = 2; defer->glm = defer->reduced = defer->staged = false;
      return MCP_OK;
    }
    if (hipStreamSynchronize(st0) != hipSuccess) { mcp_free_result(out); return fail(MCP_ERR_DEVICE, "HIP error: stream synchronize on an empty shard"); }
    if (h_vec[n_ctr0] != 0) { mcp_free_result(out); return fail(MCP_ERR_CANCELLED, "cancelled by host (on another rank of the communicator)"); }
    std::memcpy(out->counters, h_vec, (size_t)n_ctr0 * sizeof(uint64_t));
    return MCP_OK;
  }
  const bool lmg = mle && prm.lmg_active;  // random slopes / extra groupings: the general mixed-model kernel
  const bool glmm = prm.glmm_active != 0u;  // clustered GLM: random-intercept GLMM kernel (writes flags like the Mle arm)
  const bool glmm_general = glmm && prm.lmg_active && prm.lmg_n_terms > 1u;  // ... with extra groupings: joint modes, dense system
  const bool glmm_slopes = glmm && prm.lmg_active && !glmm_general;  // ... with random slopes on the primary grouping (vector modes)
  if (glmm_general && mcp::glmm_general_lds_bytes(prm) > 158u * 1024u) {
    mcp_free_result(out);
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: clustered GLM whose random effects (levels x widths over all groupings) do not fit one wave's LDS budget");
  }
  if (glmm && !glmm_general && (glmm_slopes ? mcp::glmm_slopes_lds_bytes((int)P, (int)prm.lmg_q[0], prm.max_n, prm.n_clusters_max)
                           : mcp::glmm_lds_bytes((int)P, prm.max_n, prm.n_clusters_max)) > 158u * 1024u) {
    mcp_free_result(out);
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: clustered GLM design does not fit one wave's LDS budget");
  }
  if (lmg && mcp::lmg_lds_bytes(prm) > 158u * 1024u) {
    mcp_free_result(out);
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: the random-effect structure does not fit one wave's LDS budget");
  }
  if (mle && !lmg && mcp::lmm_lds_bytes((int)P, prm.max_n, prm.n_clusters_max) > 150u * 1024u) {
    mcp_free_result(out);
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: LMM design + cluster sums do not fit one wave's LDS budget");
  }
  if (glm && !mle && !glmm && mcp::glm_lds_bytes((int)P, prm.max_n) > 150u * 1024u) {
    mcp_free_result(out);
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: GLM design (n x p f32) does not fit one wave's LDS budget");
  }

  hipStream_t st = nullptr;
  rc = ensure_device(&st, opts, defer ? &defer->arena : nullptr);
  if (rc) { mcp_free_result(out); return rc; }
  Arena& ar = defer ? defer->arena : g_arena;

  // chunking: the callback cadence of the reference is ~50 reports per run (batch.rs:450)
  uint32_t chunk = opts && opts->chunk_sims ? opts->chunk_sims : 0;
  if (!chunk) {
    chunk = staged ? (1u << 20) : (1u << 22);
    if (progress) {
      uint32_t step = n_sims / 50u;
      if (step < 16384u) step = 16384u;
      if (step < chunk) chunk = step;
    }
  }
  if (mle && !lmg) {  // the cluster-sum records between the two LMM kernels: keep one chunk's worth under 1 GiB
    const size_t per_rep = mcp::lmm_rec_fields((int)P, prm.n_clusters_max) * n_ns * sizeof(double);
    const size_t cap = ((size_t)1 << 30) / (per_rep ? per_rep : 1);
    if (cap < 64) {
      mcp_free_result(out);
      return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: LMM cluster-sum records of one replicate exceed 16 MiB");
    }
    if (chunk > cap) chunk = (uint32_t)(cap & ~(size_t)63);
  }
  // OLS on the general path or over a grid of sample sizes, narrow designs: the launch is split into a row kernel and a fit kernel (ols_kernels.hip, SPLIT) with the
  // sufficient statistics of every (replicate, grid point) between them: keep one chunk's worth under 1 GiB
  const bool ols_wide_split = !glm && !reduced && !staged && !prm.scen_active && (n_ns > 1 || P == 14u) && P >= 13u;  // ols_wide_kernel over a grid (P = 14: always)
  const bool ols_split = (!glm && !reduced && !staged && (prm.scen_active || n_ns > 1) && P <= (uint32_t)mcp::MAX_P_WAVE) || ols_wide_split;
  const size_t ols_fields = ols_wide_split ? mcp::ols_wide_rec_fields((int)P) : mcp::ols_rec_fields((int)P);
  if (ols_split) {
    const size_t per_rep = ols_fields * n_ns * sizeof(double);
    const size_t cap = ((size_t)1 << 30) / per_rep;
    if (chunk > cap) chunk = cap >= 256 ? (uint32_t)(cap & ~(size_t)255) : 256u;  // (44 fields x 256 sizes x 8 B = 90 KB per replicate at most: cap >= 11 000)
  }
  if (chunk > n_sims) chunk = n_sims;
  const uint32_t n_chunks = (n_sims + chunk - 1) / chunk;

  // a failing call may already have queued copies into `out` and kernels on the arena: drain the streams first, so that
  // nothing in flight still points at host buffers about to be freed or at device buffers the next call reuses
#define DEV_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) {                                                                                    \
      if (st) (void)hipStreamSynchronize(st);                                                                  \
      if (g_upload_stream) (void)hipStreamSynchronize(g_upload_stream);                                        \
      mcp_free_result(out);                                                                                    \
      return fail(MCP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);          \
    }                                                                                                          \
  } while (0)

  mcp::SimParams* d_prm = nullptr;
  double* d_crit = nullptr;
  uint8_t* d_levels = nullptr;
  unsigned long long* d_ctr = nullptr;
  uint8_t *d_unc = nullptr, *d_cor = nullptr, *d_conv = nullptr, *d_ov = nullptr;
  double* d_stat = nullptr;
  float* d_tile = nullptr;
  // every small per-call table travels in ONE blob: [SimParams x n_chunks | crit | crit_red | red_row | levels],
  // staged in pinned host memory so the single upload is a true async copy on the launch stream
  auto align256 = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
  const bool any_red = (prm.red_active && !glm) || glm_reduced;
  const size_t off_crit = align256(sizeof(mcp::SimParams) * n_chunks);
  const size_t off_crit_red = align256(off_crit + low.crit.size() * sizeof(double));
  const size_t off_red_row = align256(off_crit_red + (any_red ? (low.crit_red.size() + 1) * sizeof(double) : 0));
  const size_t off_levels = align256(off_red_row + (any_red ? low.red_row.size() * sizeof(uint16_t) : 0));
  const size_t blob_bytes = align256(off_levels + low.fixed_levels.size());
  uint8_t *d_blob = nullptr, *h_blob = nullptr;
  DEV_TRY(ar.get(S_PRM, blob_bytes, (void**)&d_blob));
  DEV_TRY(ar.get_pinned(0, blob_bytes, (void**)&h_blob));
  d_prm = (mcp::SimParams*)d_blob;
  d_crit = (double*)(d_blob + off_crit);
  d_levels = d_blob + off_levels;
  const uint32_t n_ctr = n_ns * layout.stride;
  // whole 128-byte lines per copy, and at least one spare word behind the counters: the reduced vector's word n_ctr
  // carries the cancel verdict through the all-reduce (zero from the memset unless a rank's callback cancelled)
  const uint32_t copy_stride = (n_ctr + 1u + 15u) & ~15u;
  const size_t ctr_words = (size_t)copy_stride * (mcp::CTR_COPIES + 1);  // copies, then the reduced vector
  DEV_TRY(ar.get(S_COUNTERS, ctr_words * sizeof(uint64_t), (void**)&d_ctr));
  uint8_t *d_ph_unc = nullptr, *d_ph_cor = nullptr, *d_fx = nullptr;
  double* d_crit_red = nullptr;
  uint16_t* d_red_row = nullptr;
  if (reduced || glm_reduced) {
    if (npc) {
      DEV_TRY(ar.get(S_PH_UNC, cells * npc + 1, (void**)&d_ph_unc));
      DEV_TRY(ar.get(S_PH_COR, cells * npc + 1, (void**)&d_ph_cor));
    }
    DEV_TRY(ar.get(S_FX, cells * (spec->n_factors ? spec->n_factors : 1) + 1, (void**)&d_fx));
    d_crit_red = (double*)(d_blob + off_crit_red);
    d_red_row = (uint16_t*)(d_blob + off_red_row);
  }
  if (per_sim || glm || reduced) {  // these kernels always write flags; a fold kernel counts them
    DEV_TRY(ar.get(S_UNC, cells * nt + 1, (void**)&d_unc));
    DEV_TRY(ar.get(S_COR, cells * nt + 1, (void**)&d_cor));
    DEV_TRY(ar.get(S_CONV, cells + 1, (void**)&d_conv));
    if (spec->report_overall) DEV_TRY(ar.get(S_OVERALL, cells + 1, (void**)&d_ov));
  }
  uint8_t *d_bh = nullptr, *d_joint = nullptr;
  double* d_lmm_rec = nullptr;
  uint32_t* d_lmm_flags = nullptr;
  const uint32_t lmm_stride = (chunk + 63u) & ~63u;
  uint16_t* d_pinned = nullptr;
  double* d_tau_hat = nullptr;
  if (lmg || glmm) DEV_TRY(ar.get(S_PINNED, (cells + 1) * sizeof(uint16_t), (void**)&d_pinned));
  if (glmm) DEV_TRY(ar.get(S_TAU_HAT, (cells + 1) * sizeof(double), (void**)&d_tau_hat));
  if (mle || glmm) {
    DEV_TRY(ar.get(S_BOUNDARY, cells + 1, (void**)&d_bh));
    DEV_TRY(ar.get(S_JOINT, cells + 1, (void**)&d_joint));
  }
  if (mle && !lmg) {
    DEV_TRY(ar.get(S_LMM_REC, mcp::lmm_rec_fields((int)P, prm.n_clusters_max) * n_ns * lmm_stride * sizeof(double), (void**)&d_lmm_rec));
    DEV_TRY(ar.get(S_LMM_FLAGS, (size_t)n_ns * lmm_stride * sizeof(uint32_t), (void**)&d_lmm_flags));
  }
  if (ols_split) DEV_TRY(ar.get(S_LMM_REC, ols_fields * n_ns * lmm_stride * sizeof(double), (void**)&d_lmm_rec));
  if (want_stat) DEV_TRY(ar.get(S_STAT, ((size_t)n_sims * nt + 1) * sizeof(double), (void**)&d_stat));
  // design tile: [block of 256 replicates][P columns: x1..x_{P-1}, y][max_n rows][256] f32
  // general path: per-replicate design records (lower factor, marginal / residual codes, per-study effects)
  const bool scen = prm.scen_active;
  const int flavor = scen ? 2 : (spec->n_factors > 0 ? 1 : 0);
  double *d_scen_L = nullptr, *d_scen_beta = nullptr, *d_t3 = nullptr, *d_scen_tau = nullptr;
  uint32_t *d_scen_meta = nullptr, *d_scen_hist = nullptr, *d_scen_perm = nullptr;
  if (scen) {
    const uint32_t K = spec->n_non_factor;
    if (spec->cluster.present)  // realised tau^2 per replicate: the primary grouping, then each extra grouping
      DEV_TRY(ar.get(S_SCEN_TAU, (size_t)chunk * (1u + spec->cluster.n_extra_groupings) * sizeof(double), (void**)&d_scen_tau));
    DEV_TRY(ar.get(S_SCEN_L, (size_t)(K * (K + 1) / 2 + 1) * chunk * sizeof(double), (void**)&d_scen_L));
    DEV_TRY(ar.get(S_SCEN_META, (size_t)chunk * sizeof(uint32_t), (void**)&d_scen_meta));
    if (low.need_beta_eff) DEV_TRY(ar.get(S_SCEN_BETA, (size_t)P * chunk * sizeof(double), (void**)&d_scen_beta));
    if (low.need_t3) {
      const std::vector<double> t3 = mcp::build_t3_table();
      DEV_TRY(ar.get(S_T3, t3.size() * sizeof(double), (void**)&d_t3));
      DEV_TRY(hipMemcpy(d_t3, t3.data(), t3.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // the lane-per-replicate fused kernel runs the replicates of a launch grouped by transform class
    // (MCP_DEBUG_NO_CLASS_ORDER=1 keeps the identity order: tests compare the two)
    static const bool no_class_order = std::getenv("MCP_DEBUG_NO_CLASS_ORDER") != nullptr;
    if (!glm && !staged && spec->n_non_factor <= 6u && !no_class_order) {
      DEV_TRY(ar.get(S_SCEN_HIST, (mcp::SCEN_CLASSES + 1u) * sizeof(uint32_t), (void**)&d_scen_hist));
      DEV_TRY(ar.get(S_SCEN_PERM, (size_t)chunk * sizeof(uint32_t), (void**)&d_scen_perm));
    }
  }
  double *d_up_normal = nullptr, *d_up_data = nullptr;
  if (spec->upload_normal_rows) {
    const size_t nb = (size_t)spec->upload_normal_rows * spec->upload_normal_cols * sizeof(double);
    DEV_TRY(ar.get(S_UPLOAD_NORMAL, nb, (void**)&d_up_normal));
    DEV_TRY(hipMemcpy(d_up_normal, spec->upload_normal, nb, hipMemcpyHostToDevice));
  }
  if (spec->n_bootstrap_frame_map) {
    const size_t nb = (size_t)spec->upload_data_rows * spec->upload_data_cols * sizeof(double);
    DEV_TRY(ar.get(S_UPLOAD_DATA, nb, (void**)&d_up_data));
    DEV_TRY(hipMemcpy(d_up_data, spec->upload_data, nb, hipMemcpyHostToDevice));
  }
  prm.upload_normal = d_up_normal;
  prm.upload_data = d_up_data;
  if (staged) DEV_TRY(ar.get(S_TILE, (size_t)((chunk + 255u) / 256u) * 256u * P * prm.max_n * sizeof(float), (void**)&d_tile));

  hipEvent_t ev_begin, ev_end;
  DEV_TRY(ar.event(0, &ev_begin));
  DEV_TRY(ar.event(1, &ev_end));
  if (!g_epoch) {  // a failed origin only costs the kernel_begin_ms / kernel_end_ms span (zeros), never the batch
    hipEvent_t ep = nullptr;
    if (hipEventCreate(&ep) == hipSuccess) {
      if (hipEventRecord(ep, st) == hipSuccess) g_epoch = ep;
      else (void)hipEventDestroy(ep);
    }
  }
  DEV_TRY(hipEventRecord(ev_begin, st));
  hipStream_t up = st;  // where the counter reset and the table upload go
  if (defer) {
    if (!g_upload_stream) DEV_TRY(hipStreamCreateWithFlags(&g_upload_stream, hipStreamNonBlocking));
    if (!defer->ev_upload) DEV_TRY(hipEventCreateWithFlags(&defer->ev_upload, hipEventDisableTiming));
    up = g_upload_stream;
  }
  DEV_TRY(hipMemsetAsync(d_ctr, 0, ctr_words * sizeof(uint64_t), up));
  std::memcpy(h_blob + off_crit, low.crit.data(), low.crit.size() * sizeof(double));
  std::memcpy(h_blob + off_levels, low.fixed_levels.data(), low.fixed_levels.size());
  if (reduced || glm_reduced) {
    std::memcpy(h_blob + off_crit_red, low.crit_red.data(), low.crit_red.size() * sizeof(double));
    std::memcpy(h_blob + off_red_row, low.red_row.data(), low.red_row.size() * sizeof(uint16_t));
  }

  prm.base_seed = base_seed;
  prm.fixed_levels = d_levels;
  prm.crit = d_crit;
  prm.unc = d_unc; prm.cor = d_cor; prm.conv = d_conv; prm.overall = d_ov; prm.stat = d_stat;
  prm.boundary = d_bh;
  prm.joint = d_joint;
  prm.ctr_joint_unc = layout.joint_unc; prm.ctr_joint_cor = layout.joint_cor;
  prm.ctr_boundary_1 = layout.boundary_1; prm.ctr_boundary_2 = layout.boundary_2;
  prm.counters = d_ctr;
  prm.ctr_copy_stride = copy_stride;
  prm.ctr_stride = layout.stride; prm.ctr_k_unc = layout.k_unc; prm.ctr_k_cor = layout.k_cor;
  prm.ctr_conv = layout.converged; prm.ctr_overall = layout.overall; prm.ctr_hist_unc = layout.hist_unc;
  prm.ctr_hist_cor = layout.hist_cor; prm.ctr_n_sims = layout.n_sims;
  prm.posthoc_unc = d_ph_unc; prm.posthoc_cor = d_ph_cor; prm.factor_excluded = d_fx;
  prm.crit_red = d_crit_red; prm.red_row = d_red_row;
  prm.ctr_posthoc_unc = layout.posthoc_unc; prm.ctr_posthoc_cor = layout.posthoc_cor;
  prm.ctr_excl_sparse = layout.excl_sparse; prm.ctr_excl_sep = layout.excl_sep;
  prm.lmm_rec = d_lmm_rec; prm.lmm_rec_flags = d_lmm_flags; prm.lmm_rec_stride = lmm_stride;
  prm.pinned = d_pinned;
  prm.tau_hat = d_tau_hat;
  prm.tile = d_tile;
  prm.tile_sims = chunk;
  prm.scen_L = d_scen_L;
  prm.scen_meta = d_scen_meta;
  prm.scen_hist = d_scen_hist;
  prm.scen_perm = d_scen_perm;
  prm.scen_beta = d_scen_beta;
  prm.scen_tau_sq = d_scen_tau;
  prm.scen_stride = chunk;
  prm.t3_table = d_t3;
  mcp::SimParams* h_prm = (mcp::SimParams*)h_blob;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const uint32_t b = c * chunk;
    std::memcpy(&h_prm[c], &prm, sizeof prm);
    h_prm[c].sim_begin = sim_offset + b;
    h_prm[c].n_sims = (n_sims - b) < chunk ? (n_sims - b) : chunk;
    h_prm[c].out_base = b;
  }
  DEV_TRY(hipMemcpyAsync(d_blob, h_blob, blob_bytes, hipMemcpyHostToDevice, up));
  if (defer) {  // the ticket's arena is its own, so this ran beside the previous batch; the launches below wait for it
    DEV_TRY(hipEventRecord(defer->ev_upload, up));
    DEV_TRY(hipStreamWaitEvent(st, defer->ev_upload, 0));
  }

  // launches, each bracketed by events on the launch stream
  const size_t ev_per_chunk = (staged || glm || reduced) ? 3 : 2;
  bool cancelled = false;
  uint32_t launched = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hipEvent_t e0, e1, e2 = nullptr;
    DEV_TRY(ar.event(2 + c * ev_per_chunk, &e0));
    DEV_TRY(ar.event(3 + c * ev_per_chunk, &e1));
    if (staged || glm || reduced) DEV_TRY(ar.event(4 + c * ev_per_chunk, &e2));
    if (scen) DEV_TRY(mcp::launch_scenario_design(d_prm + c, h_prm[c].n_sims, st, d_scen_hist));
    DEV_TRY(hipEventRecord(e0, st));
    if (glm) {
      if (glmm_general) DEV_TRY(mcp::launch_glmm_general((int)P, d_prm + c, h_prm[c].n_sims, mcp::glmm_general_lds_bytes(prm), scen, st));
      else if (glmm_slopes) DEV_TRY(mcp::launch_glmm_slopes((int)P, (int)prm.lmg_q[0], d_prm + c, h_prm[c].n_sims, prm.max_n, prm.n_clusters_max, scen, st));
      else if (glmm) DEV_TRY(mcp::launch_glmm((int)P, d_prm + c, h_prm[c].n_sims, prm.max_n, prm.n_clusters_max, scen, st));
      else if (lmg && scen) DEV_TRY(mcp::launch_lmm_general_scen((int)P, (int)prm.lmg_q[0], d_prm + c, h_prm[c].n_sims, mcp::lmg_lds_bytes(prm), st));
      else if (lmg) DEV_TRY(mcp::launch_lmm_general((int)P, (int)prm.lmg_q[0], d_prm + c, h_prm[c].n_sims, mcp::lmg_lds_bytes(prm), st));
      else if (mle) DEV_TRY(mcp::launch_lmm_sim((int)P, d_prm + c, h_prm[c].n_sims, prm.max_n, prm.n_clusters_max, spec->n_factors > 0, glm_reduced, scen, st));
      else DEV_TRY(mcp::launch_glm_sim((int)P, d_prm + c, h_prm[c].n_sims, prm.max_n, spec->n_factors > 0, glm_reduced, scen, prm.glm_family == 0u, st));
      DEV_TRY(hipEventRecord(e1, st));
      DEV_TRY(mcp::launch_fold_flags(d_prm + c, h_prm[c].n_sims, st));
      DEV_TRY(hipEventRecord(e2, st));
    } else if (reduced) {
      // cluster-level predictors under OLS, FixedClusters sizing: the first v rows park their marked values in LDS
      const uint32_t between_floats = (prm.has_cluster && prm.between_mask && prm.cluster_fixed_clusters)
                                          ? prm.cluster_v * (uint32_t)__builtin_popcount(prm.between_mask) : 0u;
      DEV_TRY(mcp::launch_ols_reduced((int)P, d_prm + c, h_prm[c].n_sims, flavor, between_floats, st));
      DEV_TRY(hipEventRecord(e1, st));
      DEV_TRY(mcp::launch_fold_flags(d_prm + c, h_prm[c].n_sims, st));
      DEV_TRY(hipEventRecord(e2, st));
    } else if (!staged) {
      DEV_TRY(mcp::launch_ols_fused((int)P, d_prm + c, h_prm[c].n_sims, n_ns == 1, flavor, n_ns, st));
      DEV_TRY(hipEventRecord(e1, st));
    } else {
      DEV_TRY(mcp::launch_ols_generate((int)P, d_prm + c, h_prm[c].n_sims, flavor, st));
      DEV_TRY(hipEventRecord(e1, st));
      DEV_TRY(mcp::launch_ols_fit((int)P, d_prm + c, h_prm[c].n_sims, n_ns == 1, st));
      DEV_TRY(hipEventRecord(e2, st));
    }
    ++launched;
    if (progress) {  // ProgressSink::report between chunks; false cancels (spec.rs:38-52)
      DEV_TRY(hipStreamSynchronize(st));
      const uint64_t done = (uint64_t)h_prm[c].out_base + h_prm[c].n_sims;
      if (!progress(done, n_sims, progress_user)) {
        cancelled = true;
        break;
      }
    }
  }
  unsigned long long* d_ctr_final = d_ctr + (size_t)copy_stride * mcp::CTR_COPIES;
  if (cancelled && pool) {
    // the other ranks are on their way to the collective: join it with the verdict word set, so that every rank
    // learns of the cancellation from the pooled payload and all of them leave together
    DEV_TRY(hipMemsetD32Async((hipDeviceptr_t)(d_ctr_final + n_ctr), 1, 1, st));
    std::string cerr;
    const int crc = mcp::comm_allreduce_u64(d_ctr_final, (size_t)n_ctr + 1u, st, cerr);
    (void)hipStreamSynchronize(st);
    mcp_free_result(out);
    if (crc) return fail(crc, cerr);
    return fail(MCP_ERR_CANCELLED, "cancelled by host");
  }
  if (cancelled) {
    (void)hipStreamSynchronize(st);
    mcp_free_result(out);
    return fail(MCP_ERR_CANCELLED, "cancelled by host");
  }

  // results
  DEV_TRY(mcp::launch_reduce_counters(d_ctr, copy_stride, n_ctr, d_ctr_final, st));
  if (pool) {  // THE collective of the path: sum over ranks, in HBM, on the launch stream (RCCL over xGMI)
    std::string cerr;
    const int crc = mcp::comm_allreduce_u64(d_ctr_final, (size_t)n_ctr + 1u, st, cerr);
    if (crc) {
      (void)hipStreamSynchronize(st);
      mcp_free_result(out);
      return fail(crc, cerr);
    }
  }
  std::vector<uint16_t> h_pinned;
  uint64_t* h_ctr = nullptr;
  DEV_TRY(ar.get_pinned(1, ((size_t)n_ctr + 1u) * sizeof(uint64_t), (void**)&h_ctr));
  DEV_TRY(hipMemcpyAsync(h_ctr, d_ctr_final, ((size_t)n_ctr + 1u) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  if (per_sim) {
    out->uncorrected = host_alloc<uint8_t>(cells * nt);
    out->corrected = host_alloc<uint8_t>(cells * nt);
    out->converged = host_alloc<uint8_t>(cells);
    out->boundary_hit = (uint8_t*)std::calloc(cells ? cells : 1, 1);         // OLS: never a boundary (batch.rs:1016)
    out->pinned_components = (uint64_t*)std::calloc(cells ? cells : 1, 8);
    out->joint_unc = (uint8_t*)std::calloc(cells ? cells : 1, 1);
    out->joint_cor = (uint8_t*)std::calloc(cells ? cells : 1, 1);
    out->tau_squared_hat = host_alloc<double>(cells);
    for (size_t i = 0; i < cells; ++i) out->tau_squared_hat[i] = NAN;
    if (spec->n_factors) out->factor_excluded = (uint8_t*)std::calloc(cells * spec->n_factors + 1, 1);
    if (npc) {
      out->posthoc_unc = (uint8_t*)std::calloc(cells * npc + 1, 1);
      out->posthoc_cor = (uint8_t*)std::calloc(cells * npc + 1, 1);
    }
    if (reduced || glm_reduced) {
      if (spec->n_factors) DEV_TRY(hipMemcpyAsync(out->factor_excluded, d_fx, cells * spec->n_factors, hipMemcpyDeviceToHost, st));
      if (npc) {
        DEV_TRY(hipMemcpyAsync(out->posthoc_unc, d_ph_unc, cells * npc, hipMemcpyDeviceToHost, st));
        DEV_TRY(hipMemcpyAsync(out->posthoc_cor, d_ph_cor, cells * npc, hipMemcpyDeviceToHost, st));
      }
    }
    DEV_TRY(hipMemcpyAsync(out->uncorrected, d_unc, cells * nt, hipMemcpyDeviceToHost, st));
    DEV_TRY(hipMemcpyAsync(out->corrected, d_cor, cells * nt, hipMemcpyDeviceToHost, st));
    DEV_TRY(hipMemcpyAsync(out->converged, d_conv, cells, hipMemcpyDeviceToHost, st));
    if (spec->report_overall) {
      out->overall = host_alloc<uint8_t>(cells);
      DEV_TRY(hipMemcpyAsync(out->overall, d_ov, cells, hipMemcpyDeviceToHost, st));
    }
    if (mle || glmm) {
      DEV_TRY(hipMemcpyAsync(out->boundary_hit, d_bh, cells, hipMemcpyDeviceToHost, st));
      DEV_TRY(hipMemcpyAsync(out->joint_unc, d_joint, cells, hipMemcpyDeviceToHost, st));
      DEV_TRY(hipMemcpyAsync(out->joint_cor, d_joint, cells, hipMemcpyDeviceToHost, st));
    }
    if (glmm) DEV_TRY(hipMemcpyAsync(out->tau_squared_hat, d_tau_hat, cells * sizeof(double), hipMemcpyDeviceToHost, st));
    if (lmg || glmm) {  // pinned-component bits travel as u16 and are widened below
      h_pinned.resize(cells);
      DEV_TRY(hipMemcpyAsync(h_pinned.data(), d_pinned, cells * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
    }
  }
  if (want_stat) {
    out->stat_capture = host_alloc<double>((size_t)n_sims * nt);
    DEV_TRY(hipMemcpyAsync(out->stat_capture, d_stat, (size_t)n_sims * nt * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  DEV_TRY(hipEventRecord(ev_end, st));
  if (defer) {  // the caller collects later: remember what the finish needs
    defer->ev_begin = ev_begin; defer->ev_end = ev_end;
    defer->h_ctr = h_ctr; defer->n_ctr = n_ctr;
    defer->launched = launched; defer->ev_per_chunk = ev_per_chunk;
    defer->glm = glm; defer->reduced = reduced; defer->staged = staged;
    return MCP_OK;
  }
  DEV_TRY(hipStreamSynchronize(st));
  if (pool && h_ctr[n_ctr] != 0) {  // some rank's progress callback cancelled: partial results are discarded everywhere
    mcp_free_result(out);
    return fail(MCP_ERR_CANCELLED, "cancelled by host (on another rank of the communicator)");
  }
  std::memcpy(out->counters, h_ctr, (size_t)n_ctr * sizeof(uint64_t));
  if ((mle || glmm) && per_sim)  // pinned_components: bit k = diagonal variance component k pinned at 0 (spec.rs:566-573)
    for (size_t i = 0; i < cells; ++i)
      out->pinned_components[i] = (lmg || glmm) ? (uint64_t)h_pinned[i] : (out->boundary_hit[i] == 1 ? 1ull : 0ull);
#undef DEV_TRY
  rc = finish_timing(ar, ev_begin, ev_end, launched, ev_per_chunk, glm, reduced, staged, out);
  if (rc) mcp_free_result(out);
  return rc;
}

}  // namespace

extern "C" {

int mcp_run_batch(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t n_sims, uint64_t base_seed,
                  const mcp_run_options* opts, mcp_progress_fn progress, void* progress_user, mcp_batch_result* out) {
  return run_batch_impl(spec, ns, n_ns, n_sims, base_seed, opts, progress, progress_user, out, nullptr);
}

int mcp_run_batch_submit(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t n_sims, uint64_t base_seed,
                         const mcp_run_options* opts, uint64_t* ticket_out) {
  if (!ticket_out) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: null ticket pointer");
  if (opts && (opts->flags & (MCP_WANT_PER_SIM | MCP_WANT_STAT_CAPTURE)))
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: mcp_run_batch_submit returns counters only (no per-sim tensors, no statistic capture)");
  if (n_sims == 0 && !(opts && (opts->flags & MCP_ALLREDUCE_COUNTERS)))  // an empty shard of a pooled call is allowed: it joins the collective
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: mcp_run_batch_submit needs n_sims > 0");
  int slot = -1;
  for (int i = 0; i < MAX_IN_FLIGHT; ++i)
    if (!g_tickets[i].busy) { slot = i; break; }
  if (slot < 0) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: too many batches in flight on this thread (collect one first)");
  Ticket& t = g_tickets[slot];
  mcp_run_options o;
  std::memset(&o, 0, sizeof o);
  if (opts) o = *opts;
  o.flags &= ~(uint32_t)MCP_WANT_PER_SIM;
  int rc = run_batch_impl(spec, ns, n_ns, n_sims, base_seed, &o, nullptr, nullptr, &t.shell, &t);
  if (rc) return rc;
  t.busy = true;
  *ticket_out = (uint64_t)slot + 1u;
  return MCP_OK;
}

int mcp_run_batch_collect(uint64_t ticket, mcp_batch_result* out) {
  if (!out) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: null result pointer");
  if (ticket == 0 || ticket > (uint64_t)MAX_IN_FLIGHT || !g_tickets[ticket - 1].busy)
    return fail(MCP_ERR_INVALID_SPEC, "invalid spec: unknown ticket (tickets are collected once, on the thread that submitted them)");
  Ticket& t = g_tickets[ticket - 1];
  t.busy = false;
  *out = t.shell;
  std::memset(&t.shell, 0, sizeof t.shell);
  hipError_t e = hipEventSynchronize(t.ev_end);
  if (e != hipSuccess) {
    mcp_free_result(out);
    return fail(MCP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e) + " at hipEventSynchronize");
  }
  std::memcpy(out->counters, t.h_ctr, (size_t)t.n_ctr * sizeof(uint64_t));
  int rc = finish_timing(t.arena, t.ev_begin, t.ev_end, t.launched, t.ev_per_chunk, t.glm, t.reduced, t.staged, out);
  if (rc) mcp_free_result(out);
  return rc;
}

int mcp_generate_design(const mcp_spec* spec, uint32_t max_n, uint32_t n_sims, uint64_t base_seed, uint64_t sim_offset,
                        float* design_out, float* outcome_out) {
  std::string err;
  const uint32_t ns1[1] = {max_n};
  int rc = mcp::validate_spec(spec, ns1, 1, err);
  if (rc) return fail(rc, err);
  mcp::SimParams prm;
  mcp::Lowered low;
  rc = mcp::lower_spec(spec, ns1, 1, prm, low, err);
  if (rc) return fail(rc, err);
  if (prm.between_mask)
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: design capture with cluster-level (between) predictors");
  if (n_sims == 0) return MCP_OK;
  hipStream_t st = nullptr;
  rc = ensure_device(&st, nullptr);
  if (rc) return rc;
  Arena& ar = g_arena;
  const uint32_t P = low.P;
  mcp::SimParams* d_prm = nullptr;
  uint8_t* d_levels = nullptr;
  float* d_tile = nullptr;
  const uint32_t n_blocks = (n_sims + 255u) / 256u;
  const uint32_t n_quads = (max_n + 3u) >> 2;
  const size_t tile_floats = (size_t)n_blocks * 256u * P * n_quads * 4u;
  HIP_TRY(ar.get(S_PRM, sizeof(mcp::SimParams), (void**)&d_prm));
  HIP_TRY(ar.get(S_LEVELS, low.fixed_levels.size(), (void**)&d_levels));
  HIP_TRY(ar.get(S_TILE, tile_floats * sizeof(float), (void**)&d_tile));
  HIP_TRY(hipMemcpyAsync(d_levels, low.fixed_levels.data(), low.fixed_levels.size(), hipMemcpyHostToDevice, st));
  prm.base_seed = base_seed;
  prm.sim_begin = sim_offset;
  prm.n_sims = n_sims;
  prm.fixed_levels = d_levels;
  prm.tile = d_tile;
  prm.tile_sims = n_sims;
  if (spec->upload_normal_rows) {
    double* d_up = nullptr;
    const size_t nb = (size_t)spec->upload_normal_rows * spec->upload_normal_cols * sizeof(double);
    HIP_TRY(ar.get(S_UPLOAD_NORMAL, nb, (void**)&d_up));
    HIP_TRY(hipMemcpy(d_up, spec->upload_normal, nb, hipMemcpyHostToDevice));
    prm.upload_normal = d_up;
  }
  if (spec->n_bootstrap_frame_map) {
    double* d_up = nullptr;
    const size_t nb = (size_t)spec->upload_data_rows * spec->upload_data_cols * sizeof(double);
    HIP_TRY(ar.get(S_UPLOAD_DATA, nb, (void**)&d_up));
    HIP_TRY(hipMemcpy(d_up, spec->upload_data, nb, hipMemcpyHostToDevice));
    prm.upload_data = d_up;
  }
  const bool scen = prm.scen_active;
  if (scen) {
    const uint32_t K = spec->n_non_factor;
    double *d_scen_L = nullptr, *d_scen_beta = nullptr, *d_t3 = nullptr, *d_scen_tau = nullptr;
    uint32_t* d_scen_meta = nullptr;
    if (spec->cluster.present)  // realised tau^2 per replicate: the primary grouping, then each extra grouping
      HIP_TRY(ar.get(S_SCEN_TAU, (size_t)n_sims * (1u + spec->cluster.n_extra_groupings) * sizeof(double), (void**)&d_scen_tau));
    prm.scen_tau_sq = d_scen_tau;
    HIP_TRY(ar.get(S_SCEN_L, (size_t)(K * (K + 1) / 2 + 1) * n_sims * sizeof(double), (void**)&d_scen_L));
    HIP_TRY(ar.get(S_SCEN_META, (size_t)n_sims * sizeof(uint32_t), (void**)&d_scen_meta));
    if (low.need_beta_eff) HIP_TRY(ar.get(S_SCEN_BETA, (size_t)P * n_sims * sizeof(double), (void**)&d_scen_beta));
    if (low.need_t3) {
      const std::vector<double> t3 = mcp::build_t3_table();
      HIP_TRY(ar.get(S_T3, t3.size() * sizeof(double), (void**)&d_t3));
      HIP_TRY(hipMemcpy(d_t3, t3.data(), t3.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    prm.scen_L = d_scen_L; prm.scen_meta = d_scen_meta; prm.scen_beta = d_scen_beta; prm.scen_stride = n_sims; prm.t3_table = d_t3;
  }
  HIP_TRY(hipMemcpyAsync(d_prm, &prm, sizeof prm, hipMemcpyHostToDevice, st));
  if (scen) HIP_TRY(mcp::launch_scenario_design(d_prm, n_sims, st));
  HIP_TRY(mcp::launch_design_capture((int)P, d_prm, n_sims, scen, st));
  std::vector<float> tile(tile_floats);
  HIP_TRY(hipMemcpyAsync(tile.data(), d_tile, tile_floats * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // tile [block][column: x1..x_{P-1}, y][row quad][256][4] -> design [sim][column 0..P-1][row], outcome [sim][row]
  auto at = [&](uint32_t s, uint32_t c, uint32_t i) {
    return tile[((((size_t)(s / 256u) * P + c) * n_quads + (i >> 2)) * 256u + (s % 256u)) * 4u + (i & 3u)];
  };
  for (uint32_t s = 0; s < n_sims; ++s) {
    float* d = design_out + (size_t)s * P * max_n;
    for (uint32_t i = 0; i < max_n; ++i) d[i] = 1.0f;
    for (uint32_t c = 1; c < P; ++c)
      for (uint32_t i = 0; i < max_n; ++i) d[(size_t)c * max_n + i] = at(s, c - 1, i);
    for (uint32_t i = 0; i < max_n; ++i) outcome_out[(size_t)s * max_n + i] = at(s, P - 1, i);
  }
  return MCP_OK;
}

int mcp_fit_design(const float* design, const float* outcome, uint32_t n_fits, uint32_t n, uint32_t p, double* beta_out,
                   double* se_out, double* t_sq_out, double* rss_out, double* sst_out, uint8_t* converged_out) {
  if (p < 2 || p > (uint32_t)mcp::MAX_P)
    return fail(MCP_ERR_UNSUPPORTED, "unsupported on the device path: design columns outside [2, 12]");
  if (n <= p) return fail(MCP_ERR_INVALID_SPEC, "invalid spec: df = N - P < 1");
  if (n_fits == 0) return MCP_OK;
  hipStream_t st = nullptr;
  int rc = ensure_device(&st, nullptr);
  if (rc) return rc;
  Arena& ar = g_arena;
  const size_t xb = (size_t)n_fits * n * p * sizeof(float), yb = (size_t)n_fits * n * sizeof(float);
  const size_t pb = (size_t)n_fits * p * sizeof(double), sb = (size_t)n_fits * sizeof(double);
  float *d_x = nullptr, *d_y = nullptr;
  double* d_out = nullptr;
  uint8_t* d_conv = nullptr;
  HIP_TRY(ar.get(S_TILE, xb, (void**)&d_x));
  HIP_TRY(ar.get(S_MISC0, yb, (void**)&d_y));
  HIP_TRY(ar.get(S_MISC1, 3 * pb + 2 * sb, (void**)&d_out));
  HIP_TRY(ar.get(S_CONV, n_fits, (void**)&d_conv));
  double* d_beta = d_out;
  double* d_se = d_out + (size_t)n_fits * p;
  double* d_tsq = d_se + (size_t)n_fits * p;
  double* d_rss = d_tsq + (size_t)n_fits * p;
  double* d_sst = d_rss + n_fits;
  HIP_TRY(hipMemcpyAsync(d_x, design, xb, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_y, outcome, yb, hipMemcpyHostToDevice, st));
  HIP_TRY(mcp::launch_ols_fit_provided((int)p, d_x, d_y, n_fits, n, d_beta, d_se, d_tsq, d_rss, d_sst, d_conv, st));
  HIP_TRY(hipMemcpyAsync(beta_out, d_beta, pb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(se_out, d_se, pb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(t_sq_out, d_tsq, pb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(rss_out, d_rss, sb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(sst_out, d_sst, sb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(converged_out, d_conv, n_fits, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MCP_OK;
}

// Debug probe of the frozen inverse-CDF kernel (rng.rs:354): out[i] = icdf(u32_to_unit_f32(first + i*stride)).
int mcp_icdf_probe(uint32_t first_word, uint32_t stride, uint32_t n, float* out) {
  if (n == 0) return MCP_OK;
  hipStream_t st = nullptr;
  int rc = ensure_device(&st, nullptr);
  if (rc) return rc;
  float* d = nullptr;
  HIP_TRY(g_arena.get(S_MISC0, (size_t)n * sizeof(float), (void**)&d));
  HIP_TRY(mcp::launch_icdf_probe(first_word, stride, n, d, st));
  HIP_TRY(hipMemcpyAsync(out, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MCP_OK;
}

}  // extern "C"
