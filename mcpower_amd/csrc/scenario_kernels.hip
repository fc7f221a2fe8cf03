// scenario_kernels.hip -- per-replicate design records of the general data-generating path.
//
// One thread per replicate draws the scenario design exactly as populate_design does
// (crates/engine-core/src/data_gen.rs:990-1093): perturbed correlation (scenarios.rs:67-122) -> PSD repair
// + Cholesky (correlation.rs:82-190) -> marginal swaps (scenarios.rs:127-157) -> residual swap (:163-186),
// all on the scenario stream SimRng(base_seed, sim_id ^ 0x5C5C...) (scenarios.rs:50), plus the per-study
// effects beta_eff on the heterogeneity stream (sim_id ^ 0x4848..., data_gen.rs:727-751).  The records
// (lower factor, marginal / residual codes, beta_eff) are written replicate-minor so the simulate kernel
// loads them coalesced.  This is cold, branchy, dynamically indexed code -- a few hundred flops per
// replicate against ~10^5 in the simulate kernel -- so it lives in its own launch.
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_scenario.hpp"

#pragma clang fp contract(off)

namespace mcp {

// Transform class of a replicate: what the lane-per-replicate simulate kernel has to EXECUTE for it -- per predictor
// "needs Phi" (any non-normal marginal) and "needs the log" (the two skewed shapes), and "residual built from the extra
// chi-square columns".  The lanes of a wave run in lockstep, so a wave pays every transform ANY of its replicates
// needs: with the frozen `realistic` preset (half of the marginals and half of the residuals swapped, independently
// per replicate) every wave of consecutive replicates pays all of them -- measured 14.4 ms per 10^5 replicates against
// 4.8 ms with no swaps.  Grouping the replicates of a launch by class lets whole waves skip what none of their lanes
// needs; the result of a replicate does not depend on the lane it runs on, the counters are sums.
__device__ __forceinline__ uint32_t scen_class(uint32_t meta, uint32_t K) {
  uint32_t c = 0;
  for (uint32_t j = 0; j < K && j < 6u; ++j) {
    const uint32_t vt = (meta >> (3u * j)) & 7u;
    c |= (vt != 0u ? 1u : 0u) << j;
    c |= ((vt == 2u || vt == 3u) ? 1u : 0u) << (6u + j);
  }
  const uint32_t rd = (meta >> 28) & 7u;
  c |= ((rd >= 1u && rd <= 3u) ? 1u : 0u) << 12;
  return c;
}

// counts -> exclusive offsets, in place (one block; SCEN_CLASSES is a multiple of the block size)
__global__ __launch_bounds__(1024) void scenario_class_scan_kernel(uint32_t* __restrict__ hist) {
  __shared__ uint32_t part[1024];
  constexpr uint32_t PER = SCEN_CLASSES / 1024u;
  uint32_t loc[PER], sum = 0;
  for (uint32_t i = 0; i < PER; ++i) {
    loc[i] = sum;
    sum += hist[threadIdx.x * PER + i];
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const uint32_t base = part[threadIdx.x] - sum;
  for (uint32_t i = 0; i < PER; ++i) hist[threadIdx.x * PER + i] = base + loc[i];
}

// lane order: replicate lsim goes to the next free slot of its class (order inside a class is whatever the atomics
// give -- no result depends on it)
__global__ __launch_bounds__(256) void scenario_class_scatter_kernel(const SimParams* __restrict__ pp) {
  const SimParams& prm = *pp;
  const uint32_t lsim = blockIdx.x * blockDim.x + threadIdx.x;
  if (lsim >= prm.n_sims) return;
  const uint32_t pos = atomicAdd(&prm.scen_hist[scen_class(prm.scen_meta[lsim], prm.K)], 1u);
  // descending class value: the costly classes (residual columns, then the log, then Phi -- the high bits) are dispatched
  // first and the cheap ones fill the tail of the launch
  prm.scen_perm[prm.n_sims - 1u - pos] = lsim;
}

__global__ __launch_bounds__(256) void scenario_design_kernel(const SimParams* __restrict__ pp) {
  const SimParams& prm = *pp;
  const uint32_t lsim = blockIdx.x * blockDim.x + threadIdx.x;
  if (lsim >= prm.n_sims) return;
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint32_t K = prm.K, P = prm.P;
  double L[MAX_K * MAX_K];  // column-major
  uint32_t vt[MAX_K];
  uint32_t rd = prm.resid_spec_dist;
  double tau_sq = prm.tau_squared;
  double extra_tau_sq[LMG_MAX_TERMS] = {0.0, prm.lmg_tau_sq[1], prm.lmg_tau_sq[2]};
  bool ok = true;
  for (uint32_t j = 0; j < K; ++j) vt[j] = prm.var_types[j];
  if (prm.scen_optimistic) {
    for (uint32_t i = 0; i < K * K; ++i) L[i] = prm.corr[i];
    ok = K == 0 || chol_colmajor(L, K);
  } else {
    SeqRng rng;
    rng.init(prm.base_seed, sim_id ^ STREAM_TAG_SCENARIO);
    if (prm.corr_noise_sd == 0.0) {
      for (uint32_t i = 0; i < K * K; ++i) L[i] = prm.corr[i];
    } else {  // scenarios.rs:67-122
      double noise[MAX_K * MAX_K];
      for (uint32_t j = 0; j < K; ++j)
        for (uint32_t i = 0; i < K; ++i) noise[j * K + i] = (double)rng.next_normal() * prm.corr_noise_sd;
      for (uint32_t j = 0; j < K; ++j)
        for (uint32_t i = 0; i < K; ++i) {
          const double sym = 0.5 * (noise[j * K + i] + noise[i * K + j]);
          const double v = prm.corr[j * K + i] + sym;
          L[j * K + i] = v < -0.8 ? -0.8 : (v > 0.8 ? 0.8 : v);
        }
      for (uint32_t i = 0; i < K; ++i) L[i * K + i] = 1.0;
    }
    ok = K == 0 || psd_repair_and_factor(L, K);
    if (prm.dist_change_prob > 0.0 && prm.n_new_dist > 0) {  // scenarios.rs:127-157
      for (uint32_t j = 0; j < K; ++j) {
        const double u = (double)rng.next_uniform();
        if (vt[j] == 0u && !prm.var_pinned[j] && u < prm.dist_change_prob) {
          uint32_t pick = (uint32_t)((double)rng.next_uniform() * (double)prm.n_new_dist);
          pick = pick > prm.n_new_dist - 1u ? prm.n_new_dist - 1u : pick;
          vt[j] = (uint32_t)prm.new_dist[pick];
        }
      }
    }
    if (prm.resid_change_prob > 0.0 && prm.n_resid_dists > 0) {  // scenarios.rs:163-186
      const double u = (double)rng.next_uniform();
      if (u < prm.resid_change_prob && prm.resid_spec_dist == 0u && !prm.residual_pinned) {
        uint32_t pick = (uint32_t)((double)rng.next_uniform() * (double)prm.n_resid_dists);
        pick = pick > prm.n_resid_dists - 1u ? prm.n_resid_dists - 1u : pick;
        rd = (uint32_t)prm.resid_dists[pick];
      }
    }
    if (prm.lme_icc_noise_sd > 0.0) {  // data_gen.rs:1069-1076: tau^2 jitter, clamped at 0
      const double jit = (double)rng.next_normal() * prm.lme_icc_noise_sd;
      tau_sq = tau_sq + jit;
      tau_sq = tau_sq > 0.0 ? tau_sq : 0.0;
      // extra groupings: one independent jitter each, after the primary's, in declaration order (:1077-1092)
      for (uint32_t t = 1; prm.lmg_active && t < prm.lmg_n_terms; ++t) {
        const double jg = (double)rng.next_normal() * prm.lme_icc_noise_sd;
        const double v = prm.lmg_tau_sq[t] + jg;
        extra_tau_sq[t] = v > 0.0 ? v : 0.0;
      }
    }
  }
  if (prm.scen_tau_sq) {
    prm.scen_tau_sq[lsim] = tau_sq;
    for (uint32_t t = 1; prm.lmg_active && t < prm.lmg_n_terms; ++t) prm.scen_tau_sq[(size_t)t * prm.scen_stride + lsim] = extra_tau_sq[t];
  }
  uint32_t meta = (rd & 7u) << 28;
  for (uint32_t j = 0; j < K; ++j) meta |= (vt[j] & 7u) << (3u * j);  // 3 bits per predictor (codes 0..7), K <= 9
  if (!ok) meta |= 0x80000000u;  // EngineError::CorrelationNotPSD for this replicate
  prm.scen_meta[lsim] = meta;
  if (prm.scen_hist) atomicAdd(&prm.scen_hist[scen_class(meta, K)], 1u);
  for (uint32_t j = 0; j < K; ++j)
    for (uint32_t t = 0; t <= j; ++t) prm.scen_L[(size_t)(j * (j + 1u) / 2u + t) * prm.scen_stride + lsim] = L[t * K + j];
  if (prm.scen_beta) {  // per-study effects, clipped toward zero, never across it (data_gen.rs:727-751)
    SeqRng hr;
    hr.init(prm.base_seed, sim_id ^ STREAM_TAG_HET);
    const double int_sd = prm.outcome_kind == 0u ? 0.0 : prm.het_total;
    for (uint32_t j = 0; j < P; ++j) {
      const double bj = prm.beta[j];
      const double sj = j == 0 ? int_sd : prm.het_total * __builtin_fabs(bj);
      const double raw = bj + (double)hr.next_normal() * sj;
      double be = raw;
      if (j != 0 && bj != 0.0) be = bj > 0.0 ? (raw > 0.0 ? raw : 0.0) : (raw < 0.0 ? raw : 0.0);
      prm.scen_beta[(size_t)j * prm.scen_stride + lsim] = be;
    }
  }
}

// hist != null: also build the class-grouped lane order of this launch (hist zeroed here, counted by the design
// kernel, scanned, then consumed by the scatter)
hipError_t launch_scenario_design(const SimParams* d_prm, uint32_t n_sims, hipStream_t st, uint32_t* d_hist) {
  if (d_hist) {
    hipError_t e = hipMemsetAsync(d_hist, 0, (SCEN_CLASSES + 1u) * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(scenario_design_kernel, dim3((n_sims + 255u) / 256u), dim3(256), 0, st, d_prm);
  if (d_hist) {
    hipLaunchKernelGGL(scenario_class_scan_kernel, dim3(1), dim3(1024), 0, st, d_hist);
    hipLaunchKernelGGL(scenario_class_scatter_kernel, dim3((n_sims + 255u) / 256u), dim3(256), 0, st, d_prm);
  }
  return hipGetLastError();
}

}  // namespace mcp
