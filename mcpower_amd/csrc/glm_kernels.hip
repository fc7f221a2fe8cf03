// glm_kernels.hip -- gfx950 kernels of the logistic-GLM (IRLS) simulate path.
//
// Mapping: ONE WAVEFRONT PER REPLICATE.  The replicate's n x p design (f32) and outcome are
// generated straight into LDS ([column][row], lanes read consecutive rows: conflict-free) and stay
// there for every IRLS pass, so the re-weighting loop never touches HBM.  Rows are spread over the
// 64 lanes; each pass ends in one butterfly all-reduce of the p(p+1)/2 + p + 1 accumulators
// (X'WX, X'Wz, deviance), after which every lane holds identical sums and runs the p x p Cholesky
// solve redundantly -- the iteration count is wave-uniform, so replicates that need more passes
// do not hold other replicates back (unlike a lane-per-replicate mapping).
//
// Reference: crates/engine-core/src/batch.rs:1562-2134 (GLM arm of run_one_sim: refit per grid N
// from the truth start beta0 = effect_sizes, Wald z^2 vs norm_ppf^2, LRT overall),
// data_gen.rs:390-398, 859-911 (uniform residual slot, Bernoulli draw).  The IRLS arithmetic
// replaces glmm 0.1.2's glm_irls_fit (out of tree): published Fisher scoring, R glm.fit's
// convergence rule |dev - dev_old| / (|dev| + 0.1) < 1e-8, <= 25 passes, |beta| > 30 -> not converged.
#ifndef MCP_GLM_WAVES
#define MCP_GLM_WAVES 1  // waves per SIMD floor for the register allocator: 2..4 change C3 by < 3% (profiles/r01_j_occupancy.txt)
#endif
#include "glm_impl.hpp"

namespace mcp {

// Fold the per-replicate flag tensors of one launch into the additive counters
// (aggregation.rs:84-284): lane per replicate, ballot + popcount, one atomic per wave per counter.
__global__ __launch_bounds__(256) void fold_flags_kernel(const SimParams* __restrict__ pp) {
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = local < prm.n_sims;
  const uint32_t row = prm.out_base + (active ? local : 0u);
  const uint32_t nt = prm.n_tests;
  const bool lane0 = (threadIdx.x & 63u) == 0u;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const size_t cell = (size_t)row * prm.n_grid + g;
    unsigned long long* ctr = prm.counters +
                            (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % CTR_COPIES) * prm.ctr_copy_stride +
                            (size_t)g * prm.ctr_stride;
    uint32_t ku = 0, kc = 0;
    for (uint32_t t = 0; t < nt; ++t) {
      const uint32_t u = active ? prm.unc[cell * nt + t] : 0u, c = active ? prm.cor[cell * nt + t] : 0u;
      ku += u != 0u;
      kc += c != 0u;
      const unsigned long long mu = __ballot(u != 0u), mc = __ballot(c != 0u);
      if (lane0) {
        if (mu) atomicAdd(ctr + prm.ctr_k_unc + t, (unsigned long long)__popcll(mu));
        if (mc) atomicAdd(ctr + prm.ctr_k_cor + t, (unsigned long long)__popcll(mc));
      }
    }
    // post-hoc contrasts count into their own slots and into the joint histogram (aggregation.rs:134-222)
    const uint32_t npc = prm.posthoc_unc ? prm.n_posthoc : 0u;
    for (uint32_t q = 0; q < npc; ++q) {
      const uint32_t u = active ? prm.posthoc_unc[cell * npc + q] : 0u, c = active ? prm.posthoc_cor[cell * npc + q] : 0u;
      ku += u != 0u;
      kc += c != 0u;
      const unsigned long long mu = __ballot(u != 0u), mc = __ballot(c != 0u);
      if (lane0) {
        if (mu) atomicAdd(ctr + prm.ctr_posthoc_unc + q, (unsigned long long)__popcll(mu));
        if (mc) atomicAdd(ctr + prm.ctr_posthoc_cor + q, (unsigned long long)__popcll(mc));
      }
    }
    for (uint32_t k = 0; k <= nt + npc; ++k) {
      const unsigned long long hu = __ballot(active && ku == k), hc = __ballot(active && kc == k);
      if (lane0) {
        if (hu) atomicAdd(ctr + prm.ctr_hist_unc + k, (unsigned long long)__popcll(hu));
        if (hc) atomicAdd(ctr + prm.ctr_hist_cor + k, (unsigned long long)__popcll(hc));
      }
    }
    if (prm.factor_excluded)  // codes: 1 = sparse level, 2 = dropped by the GLM separation fallback (aggregation.rs:244-262)
      for (uint32_t f = 0; f < prm.n_factors; ++f) {
        const uint32_t code = active ? prm.factor_excluded[cell * prm.n_factors + f] : 0u;
        const unsigned long long m1 = __ballot(code == 1u), m2 = __ballot(code == 2u);
        if (lane0) {
          if (m1) atomicAdd(ctr + prm.ctr_excl_sparse + f, (unsigned long long)__popcll(m1));
          if (m2) atomicAdd(ctr + prm.ctr_excl_sep + f, (unsigned long long)__popcll(m2));
        }
      }
    const unsigned long long mcv = __ballot(active && prm.conv[cell] != 0);
    const unsigned long long mov = __ballot(active && prm.overall && prm.overall[cell] != 0);
    const unsigned long long mac = __ballot(active);
    const unsigned long long mj = __ballot(active && prm.joint && prm.joint[cell] != 0);
    const unsigned long long mb1 = __ballot(active && prm.boundary && prm.boundary[cell] == 1);
    const unsigned long long mb2 = __ballot(active && prm.boundary && prm.boundary[cell] == 2);
    if (lane0) {
      if (mcv) atomicAdd(ctr + prm.ctr_conv, (unsigned long long)__popcll(mcv));
      if (mov) atomicAdd(ctr + prm.ctr_overall, (unsigned long long)__popcll(mov));
      if (mac) atomicAdd(ctr + prm.ctr_n_sims, (unsigned long long)__popcll(mac));
      if (mj) {
        atomicAdd(ctr + prm.ctr_joint_unc, (unsigned long long)__popcll(mj));
        atomicAdd(ctr + prm.ctr_joint_cor, (unsigned long long)__popcll(mj));
      }
      if (mb1) atomicAdd(ctr + prm.ctr_boundary_1, (unsigned long long)__popcll(mb1));
      if (mb2) atomicAdd(ctr + prm.ctr_boundary_2, (unsigned long long)__popcll(mb2));
    }
  }
}

// final[i] = sum over copies of counters[copy][i].  One wave per word: lane l adds the copies l, l + 64, ... (independent
// loads), a butterfly finishes -- the serial form (one thread walking the 256 copies of its word) was an 11 us latency
// chain on every batch's critical path (profiles/r03 C4 timeline), this one is ~2 us.
__global__ __launch_bounds__(256) void reduce_counter_copies_kernel(const unsigned long long* __restrict__ copies, uint32_t copy_stride,
                                                                    uint32_t n, unsigned long long* __restrict__ out) {
  const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long s = 0;
#pragma unroll
  for (int c = 0; c < CTR_COPIES / 64; ++c) s += copies[(size_t)(lane + 64u * (uint32_t)c) * copy_stride + i];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0u) out[i] = s;
}

hipError_t launch_reduce_counters(const unsigned long long* copies, uint32_t copy_stride, uint32_t n,
                                  unsigned long long* out, hipStream_t st) {
  static_assert(CTR_COPIES % 64 == 0, "one wave sums the copies of a word");
  // one wave per workgroup: a 256-thread workgroup needs four free wave slots on ONE CU at once and starved behind the next
  // batch's single-wave workgroups (mean residence 27 ms for microseconds of work, profiles/r03_m_r03m_lmm_slope_kernel_stats.csv)
  hipLaunchKernelGGL(reduce_counter_copies_kernel, dim3(n), dim3(64), 0, st, copies, copy_stride, n, out);
  return hipGetLastError();
}

// nt: threads sharing one replicate (64; 256 / 512 on the several-waves-per-replicate variants)
size_t glm_lds_bytes_nt(int P, uint32_t max_n, int nt) {  // staging sized for the general path's scratch column too
  const size_t n_pad = (max_n + 3u) & ~3u;
  // P > 8 (glm_wide_kernels.hip): + the column of ones and the row weights w | w z of irls_fit_wide (2 n_pad doubles);
  // nt > 64: + the cross-wave exchange of the P(P+1)/2 + P + 1 sums of a pass
  return ((size_t)P * n_pad + (size_t)4 * (P + 1) * nt) * sizeof(float) +
         (P > MAX_P_WAVE ? n_pad * sizeof(float) + (size_t)2 * n_pad * sizeof(double) : 0) +
         (nt > WAVE ? (size_t)(nt / WAVE + 1) * (P * (P + 1) / 2 + P + 1) * sizeof(double) : 0);
}
size_t glm_lds_bytes(int P, uint32_t max_n) { return glm_lds_bytes_nt(P, max_n, WAVE); }

// threads per replicate of a plain logit fit: 64 while a CU still holds two single-wave blocks per SIMD (8 blocks,
// 20 KB each); beyond that the widest block that still fits, so that the CU is not left with one or two waves
int glm_threads_per_replicate(int P, uint32_t max_n) {
  if (const char* f = std::getenv("MCP_DEBUG_GLM_THREADS")) {  // tests: force a block width on a small design
    const int nt = std::atoi(f);
    if ((nt == 256 || nt == 512) && glm_lds_bytes_nt(P, max_n, nt) <= 150u * 1024u) return nt;
    if (nt == WAVE) return WAVE;
  }
  const size_t cu = 160u * 1024u;
  if (cu / glm_lds_bytes_nt(P, max_n, WAVE) >= 8u) return WAVE;
  int best = WAVE;
  size_t best_waves = cu / glm_lds_bytes_nt(P, max_n, WAVE);
  for (int nt : {256, 512}) {
    const size_t b = glm_lds_bytes_nt(P, max_n, nt);
    if (b > 150u * 1024u) continue;
    const size_t waves = (cu / b) * (size_t)(nt / WAVE);
    if (waves > best_waves) { best = nt; best_waves = waves; }
  }
  return best;
}

template <bool FACTORS, bool RED, bool SCEN, bool LOGIT, int NT = WAVE>
static hipError_t launch_glm_f(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, hipStream_t st) {
  const size_t lds = glm_lds_bytes_nt(P, max_n, NT);
  const dim3 block(NT), grid(n_sims);
#define LG(PP)                                                                                                        \
  case PP: {                                                                                                          \
    if (lds > 48 * 1024) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)glm_sim_kernel<PP, FACTORS, RED, SCEN, LOGIT, NT>,              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return e;                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((glm_sim_kernel<PP, FACTORS, RED, SCEN, LOGIT, NT>), grid, block, lds, st, d_prm);             \
  } break;
  switch (P) {
    LG(2) LG(3) LG(4) LG(5) LG(6) LG(7) LG(8)
    default: return hipErrorInvalidValue;
  }
#undef LG
  return hipGetLastError();
}

hipError_t launch_glm_wide(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, bool factors, bool reduced,
                           bool logit, hipStream_t st);  // glm_wide_kernels.hip: P = 9..16

hipError_t launch_glm_sim(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, bool factors, bool reduced,
                          bool scen, bool logit, hipStream_t st) {
  if (P > MAX_P_WAVE) return scen ? hipErrorInvalidValue : launch_glm_wide(P, d_prm, n_sims, max_n, factors, reduced, logit, st);
  const int nt = scen ? WAVE : glm_threads_per_replicate(P, max_n);  // large designs: several waves share one replicate
  if (!logit) {  // probit / Poisson: one run-time-family variant (the superset kernel)
    if (scen) return launch_glm_f<true, true, true, false>(P, d_prm, n_sims, max_n, st);
    return nt > WAVE ? launch_glm_f<true, true, false, false, 256>(P, d_prm, n_sims, max_n, st)
                     : launch_glm_f<true, true, false, false>(P, d_prm, n_sims, max_n, st);
  }
  if (scen) return launch_glm_f<true, true, true, true>(P, d_prm, n_sims, max_n, st);
  if (reduced) return nt > WAVE ? launch_glm_f<true, true, false, true, 256>(P, d_prm, n_sims, max_n, st)
                                : launch_glm_f<true, true, false, true>(P, d_prm, n_sims, max_n, st);
  if (factors) return nt > WAVE ? launch_glm_f<true, false, false, true, 256>(P, d_prm, n_sims, max_n, st)
                                : launch_glm_f<true, false, false, true>(P, d_prm, n_sims, max_n, st);
  if (nt == 512) return launch_glm_f<false, false, false, true, 512>(P, d_prm, n_sims, max_n, st);
  if (nt == 256) return launch_glm_f<false, false, false, true, 256>(P, d_prm, n_sims, max_n, st);
  return launch_glm_f<false, false, false, true>(P, d_prm, n_sims, max_n, st);
}

hipError_t launch_fold_flags(const SimParams* d_prm, uint32_t n_sims, hipStream_t st) {
  // 64-thread workgroups: the fold of batch i competes for wave slots with the main kernel of batch i + 1 (its own
  // stream), whose single-wave workgroups refill every slot the moment it frees -- a 256-thread workgroup needs four
  // free slots on ONE CU at once and starved behind them (mean residence 45 ms for 4 us of work,
  // profiles/r02_c_lmm_slope_kernel_stats.csv)
  hipLaunchKernelGGL(fold_flags_kernel, dim3((n_sims + 63u) / 64u), dim3(64), 0, st, d_prm);
  return hipGetLastError();
}

}  // namespace mcp
