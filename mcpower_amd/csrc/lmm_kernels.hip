// lmm_kernels.hip -- gfx950 kernel of the random-intercept LMM (REML) simulate path.
//
// A random intercept's profiled REML deviance depends on the data only through per-cluster sums
// (n_c, sum x_c, sum y_c) and the global X'X, X'y, y'y (scripts/lme_handcompute_theta1.py:55-93).
// The path is therefore split in two launches with opposite shapes:
//   1. lmm_prepare_kernel -- ONE WAVEFRONT PER REPLICATE: the n x p design is generated into LDS (rows spread
//      over the lanes, like the GLM kernel), then for every grid N one pass (lane = cluster) forms the cluster
//      sums v_c = (n_c, Sx_c, Sy_c).  The deviance needs them only through sum_c s(n_c) v_c v_c', and the
//      engine's cluster layouts produce at most TWO distinct cluster sizes (i % n_clusters, or full blocks plus
//      one partial), so the wave reduces M_k = sum_{c in size class k} v_c v_c' for k = 0, 1 and writes
//      2 (p+1)(p+2)/2 + 4 + p(p+1)/2 + p + 1 doubles per fit, replicate-minor, plus the sparse-factor flags:
//      an objective evaluation no longer depends on the number of clusters at all.
//   2. lmm_optimise_kernel -- ONE LANE PER REPLICATE: Brent's minimiser, the p x p Cholesky, Wald / joint
//      statistics and the decisions are a few hundred dependent f64 operations per objective evaluation with
//      no parallelism inside a replicate worth a wave; 64 replicates per wave run them without a single
//      cross-lane reduction, reading the class matrices back coalesced ([field][grid][replicate]).
// The first version ran the optimiser wave-per-replicate (lane = cluster + a butterfly all-reduce of
// p(p+1)/2 + p + 2 values per evaluation): 30 of 64 lanes busy, ~70 dependent shuffles per evaluation, three
// waves per SIMD -- latency-bound at 3 % of the vector issue rate (profiles/r01_c_*).
//
// Reference: crates/engine-core/src/batch.rs:1214-1560 (Mle arm), data_gen.rs:146-161 (cluster
// random-intercept draws on the sequential stream before the row loop), :811-857 (u_re added on the
// latent scale), engine-contract/src/generation.rs:125-158 (row -> cluster).  The optimiser replaces
// glmm 0.1.2 fit_on + bobyqa 0.1.1 (out of tree) with Brent's method on theta in [0, 32].
#ifndef MCP_LMM_WAVES
#define MCP_LMM_WAVES 1  // waves per SIMD floor for the allocator: no gain from 2..4 on MI355X (profiles/r01_j_occupancy.txt)
#endif
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"

#pragma clang fp contract(off)

namespace mcp {

constexpr int LWAVE = 64;
constexpr double THETA_MAX = 32.0;
constexpr double THETA_BLIND = 1.0;
constexpr double THETA_TOL = 1e-9;
constexpr double THETA_ZERO = 1e-6;
constexpr int BRENT_MAX_ITERS = 100;
constexpr double LMM_NEAR_ZERO = 1e-30;

__device__ __forceinline__ double lwave_sum(double v) {
#pragma unroll
  for (int m = 1; m < LWAVE; m <<= 1) v += __shfl_xor(v, m, LWAVE);
  return v;
}

// ---- record layout in HBM: rec[(field * n_grid + g) * stride + replicate] ----
// fields: M_0, M_1 (lower triangles of sum v v', v = (n_c, Sx_1..Sx_{P-1}, Sy)), then {size_0, size_1, count_0,
// count_1}, then X'X lower (P(P+1)/2), X'y (P), y'y (1)
template <int P>
__host__ __device__ constexpr uint32_t lmm_m_fields() { return (P + 1) * (P + 2) / 2; }
template <int P>
__host__ __device__ constexpr uint32_t lmm_total_fields() { return 2 * lmm_m_fields<P>() + 4 + P * (P + 1) / 2 + P + 1; }

__device__ __forceinline__ size_t rec_at(const SimParams& prm, uint32_t field, uint32_t g, uint32_t rep) {
  return ((size_t)field * prm.n_grid + g) * prm.lmm_rec_stride + rep;
}

// LDS layout per block (one wave): design [P columns: x1..x_{P-1}, y][n_pad] f32 | cluster_u [ncl_pad] f32 |
// staging [4(P+1)][64] f32
// RED: exclusion flags recounted from rows [0, N) before every fit (batch.rs:1236-1250).
// SCEN: general data path (scenario design records, non-normal marginals / residuals, per-study effects,
// heteroskedastic residuals) plus the LME knobs: random-effect shape and per-replicate tau^2 (data_gen.rs:146-161).
template <int P, bool FACTORS, bool RED, bool SCEN>
// P <= 3 on the plain data path (BASELINE configs[3]: 30 clusters x 20, P = 3) is held to 128 registers: a CU then holds the
// 14 single-wave workgroups its LDS admits (11.4 KB each) instead of 12, and 10^4 replicates take 3 rounds of resident
// waves instead of 4; the wider / general-path variants keep the default budget (they need 200+)
__global__ __launch_bounds__(LWAVE, (P <= 3 && !SCEN) ? 4 : 1) void lmm_prepare_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ float lds_f[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x;
  const uint32_t lsim = blockIdx.x;
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t max_n = prm.max_n;
  const uint32_t n_pad = (max_n + 3u) & ~3u;
  const uint32_t ncl_max = prm.n_clusters_max;
  const uint32_t ncl_pad = (ncl_max + 1u) & ~1u;
  const uint32_t cv = prm.cluster_v;
  float* design = lds_f;                          // P * n_pad
  float* cluster_u = design + (size_t)P * n_pad;  // ncl_pad
  float* st = cluster_u + ncl_pad + lane;         // staging, stride 64

  // ---- cluster random intercepts: draw c is word c of the sequential stream (data_gen.rs:146-161) ----
  double tau_rep = prm.tau;
  if (SCEN && prm.scen_tau_sq) {
    const double t2 = prm.scen_tau_sq[lsim];
    tau_rep = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
  }
  for (uint32_t c = lane; c < ncl_max; c += LWAVE) {
    if (SCEN) {
      cluster_u[c] = (float)(draw_cluster_effect(prm, k0, k1, c) * tau_rep);
    } else {
      const float z = norm_inv_cdf_f32(u32_to_unit_f32(sequential_word(k0, k1, (uint64_t)c)));
      cluster_u[c] = (float)((double)z * prm.tau);
    }
  }
  __syncthreads();

  // ---- generate the replicate into LDS ----
  RowCtx<P> cx;
  row_ctx_init<P, FACTORS>(prm, cx);
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);
  const uint32_t bmask = prm.between_mask;
  for (uint32_t ctr = lane; 4u * ctr < max_n; ctr += LWAVE) {
    stage_group<P, LWAVE, 0, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rr = 4u * ctr + r;
      if (rr < max_n) {
        const uint32_t cid = prm.cluster_fixed_clusters ? rr % cv : rr / cv;
        const double u_re = (double)cluster_u[cid < ncl_max ? cid : 0u];
        float x[P], y;
        const float e = build_x<P, FACTORS, LWAVE, SCEN>(prm, cx, k0, k1, st, r, rr, x, &sl);
        if (bmask == 0u) finish_row<P, 0, SCEN>(prm, cx, x, e, y, u_re, &sl);
        else y = e;  // the residual waits in the outcome slot until the cluster-level columns are broadcast
#pragma unroll
        for (int c = 1; c < P; ++c) design[(size_t)(c - 1) * n_pad + rr] = x[c];
        design[(size_t)(P - 1) * n_pad + rr] = y;
      }
    }
  }
  __syncthreads();
  if (bmask != 0u) {
    // cluster-level predictors (data_gen.rs:606-636): every row takes the marked columns from its cluster's
    // representative row (row i % n_clusters, or the first row of the block), then interactions and the outcome
    // are formed from the broadcast values.  Representative rows keep their own values, so the copies below
    // read final data whatever order the lanes run in.
    for (uint32_t rr = lane; rr < max_n; rr += LWAVE) {
      const uint32_t rep = prm.cluster_fixed_clusters ? rr % cv : rr - (rr % cv);
      float x[P];
      x[0] = 1.0f;
#pragma unroll
      for (int c = 1; c < P; ++c) {
        const uint32_t src = ((bmask >> c) & 1u) ? rep : rr;
        x[c] = design[(size_t)(c - 1) * n_pad + src];
      }
      rebuild_interactions<P>(cx, x);
      const float e = design[(size_t)(P - 1) * n_pad + rr];
      const uint32_t cid = prm.cluster_fixed_clusters ? rr % cv : rr / cv;
      float y;
      finish_row<P, 0, SCEN>(prm, cx, x, e, y, (double)cluster_u[cid < ncl_max ? cid : 0u], &sl);
      __builtin_amdgcn_wave_barrier();  // all reads of this sweep's source columns happen before any overwrite
#pragma unroll
      for (int c = 1; c < P; ++c)
        if (((bmask >> c) & 1u) == 0u && ((cx.inter_any >> c) & 1u)) design[(size_t)(c - 1) * n_pad + rr] = x[c];
      design[(size_t)(P - 1) * n_pad + rr] = y;
    }
    // marked columns are rewritten last: until here every row still saw the representatives' originals
    __syncthreads();
    for (uint32_t rr = lane; rr < max_n; rr += LWAVE) {
      const uint32_t rep = prm.cluster_fixed_clusters ? rr % cv : rr - (rr % cv);
#pragma unroll
      for (int c = 1; c < P; ++c)
        if ((bmask >> c) & 1u) {
          const float v = design[(size_t)(c - 1) * n_pad + rep];
          __builtin_amdgcn_wave_barrier();
          design[(size_t)(c - 1) * n_pad + rr] = v;
        }
    }
    __syncthreads();
  }

  double* rec = prm.lmm_rec;
  constexpr uint32_t MF = lmm_m_fields<P>();
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    const uint32_t ncl = prm.cluster_fixed_clusters ? cv : (n + cv - 1u) / cv;
    // cluster sizes are analytic: i % ncl gives ceil / floor(n / ncl); blocks of cv give cv and one partial block
    auto size_of = [&](uint32_t c) -> uint32_t {
      if (prm.cluster_fixed_clusters) return n > c ? (n - 1u - c) / cv + 1u : 0u;
      const uint32_t lo = c * cv;
      return lo >= n ? 0u : (n - lo < cv ? n - lo : cv);
    };
    const uint32_t size0 = size_of(0u), size1 = size_of(ncl - 1u);  // the largest and the smallest
    // ---- per-cluster sums (lane = cluster), class matrices, global X'X, X'y, y'y ----
    double xtx[P * (P + 1) / 2], xty[P], yty = 0.0;
    double M0[MF], M1[MF];
    double cnt0 = 0.0, cnt1 = 0.0;
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; ++i) xtx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) xty[i] = 0.0;
#pragma unroll
    for (uint32_t i = 0; i < MF; ++i) {
      M0[i] = 0.0;
      M1[i] = 0.0;
    }
    // few clusters leave lanes idle (30 of 64 at BASELINE configs[3]): with <= 32 clusters the lanes l and l + 32 share
    // cluster l, each takes every other row of it, and one exchange leaves the cluster's sums v_c in both
    const uint32_t halves = ncl <= 32u ? 2u : 1u;
    const uint32_t half = halves == 2u ? (lane >> 5) : 0u;
    for (uint32_t c = halves == 2u ? (lane & 31u) : lane; c < ncl; c += LWAVE) {
      double v[P + 1];  // (n_c, Sx_1..Sx_{P-1}, Sy)
#pragma unroll
      for (int a = 0; a <= P; ++a) v[a] = 0.0;
      const uint32_t step1 = prm.cluster_fixed_clusters ? cv : 1u;
      const uint32_t first = (prm.cluster_fixed_clusters ? c : c * cv) + half * step1;
      const uint32_t step = step1 * halves;
      const uint32_t last = prm.cluster_fixed_clusters ? n : ((c + 1u) * cv < n ? (c + 1u) * cv : n);
      for (uint32_t i = first; i < last; i += step) {
        double xd[P];
        xd[0] = 1.0;
#pragma unroll
        for (int a = 1; a < P; ++a) xd[a] = (double)design[(size_t)(a - 1) * n_pad + i];
        const double yi = (double)design[(size_t)(P - 1) * n_pad + i];
        v[0] += 1.0;
        v[P] += yi;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < P; ++a) {
          if (a > 0) v[a] += xd[a];
#pragma unroll
          for (int q = 0; q <= a; ++q) {
            xtx[idx] = __builtin_fma(xd[a], xd[q], xtx[idx]);
            ++idx;
          }
          xty[a] = __builtin_fma(xd[a], yi, xty[a]);
        }
        yty = __builtin_fma(yi, yi, yty);
      }
      if (halves == 2u) {
#pragma unroll
        for (int a = 0; a <= P; ++a) v[a] += __shfl_xor(v[a], 32, LWAVE);  // a + b == b + a: both halves hold the same sums
      }
      const uint32_t sz = size_of(c);
      if (sz != 0u && half == 0u) {
        const bool k0c = sz == size0;
        cnt0 += k0c ? 1.0 : 0.0;
        cnt1 += k0c ? 0.0 : 1.0;
        int idx = 0;
#pragma unroll
        for (int a = 0; a <= P; ++a)
#pragma unroll
          for (int q = 0; q <= a; ++q) {
            const double pr = v[a] * v[q];
            M0[idx] += k0c ? pr : 0.0;
            M1[idx] += k0c ? 0.0 : pr;
            ++idx;
          }
      }
    }
    // The record (order: M0 | M1 | size0 size1 cnt0 cnt1 | X'X | X'y | y'y) is reduced over the wave through the staging
    // area (dead once the rows exist): CH = 2 (P + 1) fields at a time, lane l parks its partial of field k at [k][l];
    // lane (k = l mod 8 .., part = l / CH ..) then adds a run of the 64 partials of field k in lane order and a short
    // butterfly over the parts finishes -- ~35 instructions per chunk where 6-step butterflies of every field took
    // 18 each (a third of this kernel's non-RNG instructions at P = 3).  Fixed summation order: deterministic.
    {
      constexpr int NV = 2 * (int)MF + 2 + P * (P + 1) / 2 + P + 1;  // reduced fields
      double vals[NV];
      {
        int o = 0;
#pragma unroll
        for (uint32_t i = 0; i < MF; ++i) vals[o++] = M0[i];
#pragma unroll
        for (uint32_t i = 0; i < MF; ++i) vals[o++] = M1[i];
        vals[o++] = cnt0;
        vals[o++] = cnt1;
#pragma unroll
        for (int i = 0; i < P * (P + 1) / 2; ++i) vals[o++] = xtx[i];
#pragma unroll
        for (int i = 0; i < P; ++i) vals[o++] = xty[i];
        vals[o++] = yty;
      }
      // record slot of reduced field j: the two uniform size fields sit behind M0 | M1
      auto slot_of = [&](uint32_t j) -> uint32_t { return j < 2u * MF ? j : j + 2u; };
      constexpr int CH = 8;                       // fields per chunk: 8 doubles per lane fit the staging area from P = 3 on
      static_assert(4 * (P + 1) * (int)sizeof(float) >= CH * (int)sizeof(double) || P < 3, "staging area too small for a chunk");
      constexpr int CHP = P >= 3 ? CH : 4;        // P = 2: 4 (P + 1) floats = 6 doubles per lane
      constexpr int PARTS = LWAVE / CHP, RUN = LWAVE / PARTS;
      double* sbuf = reinterpret_cast<double*>(cluster_u + ncl_pad);  // the staging area (8-byte aligned: ncl_pad is even)
#pragma unroll
      for (int c0 = 0; c0 < NV; c0 += CHP) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CHP; ++k)
          if (c0 + k < NV) sbuf[k * LWAVE + lane] = vals[c0 + k];
        __syncthreads();
        const uint32_t k = lane % (uint32_t)CHP, part = lane / (uint32_t)CHP;
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < RUN; ++j) t += sbuf[k * LWAVE + part * RUN + j];
#pragma unroll
        for (int m = CHP; m < LWAVE; m <<= 1) t += __shfl_xor(t, m, LWAVE);
        if (lane < (uint32_t)CHP && c0 + (int)lane < NV) rec[rec_at(prm, slot_of((uint32_t)c0 + lane), g, lsim)] = t;
      }
      if (lane == 0) {
        rec[rec_at(prm, 2u * MF, g, lsim)] = (double)size0;
        rec[rec_at(prm, 2u * MF + 1u, g, lsim)] = (double)size1;
      }
      __syncthreads();
    }
    uint32_t flag_bits = 0;
    if (RED && prm.min_level_count) {  // update_factor_exclusions (batch.rs:2451-2500), rows [0, n)
#pragma unroll
      for (int f = 0; f < MAX_FACTORS; ++f) {
        if ((uint32_t)f < prm.n_factors) {
          uint32_t mn = 0xffffffffu;
          for (uint32_t l = 0; l < prm.f_levels[f]; ++l) {
            uint32_t cnt = 0;
            for (uint32_t i = lane; i < n; i += LWAVE) {
              uint32_t lv = 0;
#pragma unroll
              for (int c = 1; c < P; ++c)
                if (prm.col_factor[c] == (uint32_t)f && design[(size_t)(c - 1) * n_pad + i] == 1.0f) lv = prm.col_level[c];
              cnt += lv == l ? 1u : 0u;
            }
#pragma unroll
            for (int m = 1; m < LWAVE; m <<= 1) cnt += __shfl_xor(cnt, m, LWAVE);
            mn = cnt < mn ? cnt : mn;
          }
          if (mn < prm.min_level_count) flag_bits |= 1u << f;
        }
      }
    }
    if (lane == 0) prm.lmm_rec_flags[(size_t)g * prm.lmm_rec_stride + lsim] = flag_bits;
  }
}

template <int P>
struct LmmEval {
  bool ok;
  double dev, sigma_sq;
  double beta[P];
  double Lm[P][P];
  double invd[P];
};

// dev(theta) and the REML solution at theta for ONE replicate held by this lane; the cluster sums stream from
// HBM, coalesced across the wave's 64 replicates.  `keep` selects the fitted columns (identity rows/columns of
// X'V^-1X at the dropped ones, zero right-hand side), df = n - popcount(keep): batch.rs:1252-1394.
// dev(theta) and the REML solution at theta for ONE replicate held by this lane.  sum_c s(n_c) v_c v_c' collapses
// to s(size_0) M_0 + s(size_1) M_1 and sum_c log(1 + theta^2 n_c) to count_0 log(..size_0) + count_1 log(..size_1).
// `keep` selects the fitted columns (identity rows/columns of X'V^-1X at the dropped ones, zero right-hand
// side), df = n - popcount(keep): batch.rs:1252-1394.
template <int P, bool MASKED>
__device__ __forceinline__ void lmm_eval_lane(const double (&M0)[(P + 1) * (P + 2) / 2], const double (&M1)[(P + 1) * (P + 2) / 2],
                                              double size0, double size1, double cnt0, double cnt1,
                                              const double (&xtx)[P * (P + 1) / 2], const double (&xty)[P], double yty,
                                              uint32_t n, double theta, uint32_t keep, LmmEval<P>& e) {
  const double th2 = theta * theta;
  const double s0 = cnt0 > 0.0 ? th2 / (1.0 + th2 * size0) : 0.0;
  const double s1 = cnt1 > 0.0 ? th2 / (1.0 + th2 * size1) : 0.0;
  double dA[P * (P + 1) / 2], db[P];
  {
    int idx = 0, mi = 0;
#pragma unroll
    for (int a = 0; a <= P; ++a)
#pragma unroll
      for (int q = 0; q <= a; ++q) {
        const double v = __builtin_fma(s1, M1[mi], s0 * M0[mi]);
        if (a < P) dA[idx++] = v;
        else if (q < P) db[q] = v;
        ++mi;
      }
  }
  const double dyy = __builtin_fma(s1, M1[(P + 1) * (P + 2) / 2 - 1], s0 * M0[(P + 1) * (P + 2) / 2 - 1]);
  const double ldv = (cnt0 > 0.0 ? cnt0 * log(1.0 + th2 * size0) : 0.0) + (cnt1 > 0.0 ? cnt1 * log(1.0 + th2 * size1) : 0.0);
  double A[P][P], b[P];
  {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < P; ++a) {
#pragma unroll
      for (int q = 0; q <= a; ++q) {
        const double v = xtx[idx] - dA[idx];
        A[a][q] = (!MASKED || (((keep >> a) & 1u) && ((keep >> q) & 1u))) ? v : (a == q ? 1.0 : 0.0);
        ++idx;
      }
      const double bv = xty[a] - db[a];
      b[a] = (!MASKED || ((keep >> a) & 1u)) ? bv : 0.0;
    }
  }
  const double yy = yty - dyy;
  const double logdetv = ldv;
  e.ok = false;
  e.dev = __builtin_inf();
  bool ok = true;
  double logdetA = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double dd = A[j][j];
#pragma unroll
    for (int t = 0; t < j; ++t) dd = __builtin_fma(-e.Lm[j][t], e.Lm[j][t], dd);
    ok = ok && (dd > 0.0) && (dd < __builtin_inf());
    const double ljj = __builtin_sqrt(dd);
    e.Lm[j][j] = ljj;
    e.invd[j] = 1.0 / ljj;
    logdetA += 2.0 * log(ljj);
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int t = 0; t < j; ++t) v = __builtin_fma(-e.Lm[i][t], e.Lm[j][t], v);
      e.Lm[i][j] = v * e.invd[j];
    }
  }
  if (!ok) return;
  double zf[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    double v = b[i];
#pragma unroll
    for (int t = 0; t < i; ++t) v = __builtin_fma(-e.Lm[i][t], zf[t], v);
    zf[i] = v * e.invd[i];
  }
#pragma unroll
  for (int i = P - 1; i >= 0; --i) {
    double v = zf[i];
#pragma unroll
    for (int t = i + 1; t < P; ++t) v = __builtin_fma(-e.Lm[t][i], e.beta[t], v);
    e.beta[i] = v * e.invd[i];
  }
  double bxy = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) bxy = __builtin_fma(e.beta[i], b[i], bxy);
  const double df = (double)n - (MASKED ? (double)__popc(keep & ((1u << P) - 1u)) : (double)P);
  e.sigma_sq = (yy - bxy) / df;
  if (!(e.sigma_sq > 0.0) || !(e.sigma_sq < __builtin_inf())) return;
  e.dev = logdetv + logdetA + df * log(e.sigma_sq);
  e.ok = e.dev < __builtin_inf() && e.dev > -__builtin_inf();
}

// ONE LANE PER REPLICATE: grid-sequential REML fits (theta carried along the grid, batch.rs:1323-1332,
// :1423-1430), Wald z^2, joint Wald chi2, decisions.  RED: a reduced round fits the kept columns, never feeds the
// theta carry (:1414-1428) and runs the joint Wald over the kept targets against chi2(k_red) (:1456-1468).
template <int P, bool RED>
__global__ __launch_bounds__(64, MCP_LMM_WAVES) void lmm_optimise_kernel(const SimParams* __restrict__ pp) {
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = local < prm.n_sims;
  const uint32_t lsim = active ? local : (prm.n_sims - 1u);
  const uint32_t out_row = prm.out_base + lsim;
  constexpr uint32_t MF = lmm_m_fields<P>();
  constexpr uint32_t f_fixed = 2 * MF + 4;
  const double* __restrict__ rec = prm.lmm_rec;
  const uint32_t nt = prm.n_tests;
  bool has_prev = false;
  double theta_carry = 0.0;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    double M0[MF], M1[MF];
#pragma unroll
    for (uint32_t i = 0; i < MF; ++i) {
      M0[i] = rec[rec_at(prm, i, g, lsim)];
      M1[i] = rec[rec_at(prm, MF + i, g, lsim)];
    }
    const double size0 = rec[rec_at(prm, 2 * MF, g, lsim)], size1 = rec[rec_at(prm, 2 * MF + 1, g, lsim)];
    const double cnt0 = rec[rec_at(prm, 2 * MF + 2, g, lsim)], cnt1 = rec[rec_at(prm, 2 * MF + 3, g, lsim)];
    double xtx[P * (P + 1) / 2], xty[P];
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; ++i) xtx[i] = rec[rec_at(prm, f_fixed + i, g, lsim)];
#pragma unroll
    for (int i = 0; i < P; ++i) xty[i] = rec[rec_at(prm, f_fixed + P * (P + 1) / 2 + i, g, lsim)];
    const double yty = rec[rec_at(prm, f_fixed + P * (P + 1) / 2 + P, g, lsim)];
    const uint32_t rflags = prm.lmm_rec_flags[(size_t)g * prm.lmm_rec_stride + lsim];
    const uint32_t flag_bits = RED ? (rflags & 15u) : 0u;
    const uint32_t keep = RED ? prm.keep_mask[flag_bits] : ((1u << P) - 1u);
    const bool reduced_round = RED && keep != ((1u << P) - 1u);

    // ---- Brent's minimiser of dev(theta) on [0, THETA_MAX] (fmin), first probe at the start value ----
    double start = has_prev ? theta_carry : (prm.truth_start ? prm.tau : THETA_BLIND);
    start = start < 0.01 ? 0.01 : start;  // THETA_TRUTH_FLOOR (batch.rs:1316-1321)
    LmmEval<P> ev;
    const double CGOLD = 0.3819660112501051;
    double a = 0.0, b = THETA_MAX;
    double x = start;
    x = (x > a + 1e-3) ? x : a + 1e-3;
    x = (x < b - 1e-3) ? x : b - 1e-3;
    double w = x, v = x, e = 0.0, dstep = 0.0;
    lmm_eval_lane<P, RED>(M0, M1, size0, size1, cnt0, cnt1, xtx, xty, yty, n, x, keep, ev);
    double fx = ev.dev, fw = fx, fv = fx;
    for (int it = 0; it < BRENT_MAX_ITERS; ++it) {
      const double xm = 0.5 * (a + b);
      const double tol1 = THETA_TOL * __builtin_fabs(x) + 1e-10, tol2 = 2.0 * tol1;
      if (__builtin_fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
      bool golden = true;
      if (__builtin_fabs(e) > tol1) {
        const double r = (x - w) * (fx - fv);
        double q = (x - v) * (fx - fw);
        double pq = (x - v) * q - (x - w) * r;
        q = 2.0 * (q - r);
        if (q > 0.0) pq = -pq;
        q = __builtin_fabs(q);
        const double etemp = e;
        e = dstep;
        if (!(__builtin_fabs(pq) >= __builtin_fabs(0.5 * q * etemp) || pq <= q * (a - x) || pq >= q * (b - x))) {
          dstep = pq / q;
          const double u = x + dstep;
          if (u - a < tol2 || b - u < tol2) dstep = (xm - x >= 0.0) ? tol1 : -tol1;
          golden = false;
        }
      }
      if (golden) {
        e = (x >= xm) ? a - x : b - x;
        dstep = CGOLD * e;
      }
      const double u = __builtin_fabs(dstep) >= tol1 ? x + dstep : x + (dstep >= 0.0 ? tol1 : -tol1);
      lmm_eval_lane<P, RED>(M0, M1, size0, size1, cnt0, cnt1, xtx, xty, yty, n, u, keep, ev);
      const double fu = ev.dev;
      if (fu <= fx) {
        if (u >= x) a = x; else b = x;
        v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
      } else {
        if (u < x) a = u; else b = u;
        if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
        else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
      }
    }
    // the interior search never evaluates theta = 0: compare with the boundary value
    double th = x;
    lmm_eval_lane<P, RED>(M0, M1, size0, size1, cnt0, cnt1, xtx, xty, yty, n, 0.0, keep, ev);
    const bool at0_ok = ev.ok;
    const double dev0 = ev.dev;
    const bool interior_ok = fx < __builtin_inf() && fx > -__builtin_inf();
    if (at0_ok && (!interior_ok || dev0 <= fx || th < THETA_ZERO)) {
      th = 0.0;  // ev already holds the theta = 0 solution
    } else {
      lmm_eval_lane<P, RED>(M0, M1, size0, size1, cnt0, cnt1, xtx, xty, yty, n, th, keep, ev);
    }
    bool conv = ev.ok;
    const uint32_t bh = !ev.ok ? 0u : (th < THETA_ZERO ? 1u : (th > THETA_MAX - 1e-3 ? 2u : 0u));
    if (bh == 2u) conv = false;
    if (conv && !reduced_round) {
      theta_carry = th;
      has_prev = true;
    }
    // ---- Wald z^2 (cov = sigma^2 (X'V^-1X)^-1), joint Wald chi2 over the targets, decisions ----
    const double* crow = prm.crit + (size_t)g * prm.crit_stride;
    double tsq[MAX_T_WAVE];
    double U[MAX_T_WAVE][P];  // L^-1 c_t
    uint32_t tkeep = 0;  // targets whose column is fitted
#pragma unroll
    for (int t = 0; t < MAX_T_WAVE; ++t)
      if ((uint32_t)t < nt && (!RED || ((keep >> prm.test_pos[t]) & 1u))) tkeep |= 1u << t;
#pragma unroll
    for (int t = 0; t < MAX_T_WAVE; ++t) {
      tsq[t] = __builtin_nan("");
#pragma unroll
      for (int i = 0; i < P; ++i) U[t][i] = 0.0;
      if (((tkeep >> t) & 1u) && conv) {
        const double* c = prm.contrast + t * MAX_P;
        double est = 0.0, nq = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
          est = __builtin_fma(c[i], ev.beta[i], est);
          double acc = c[i];
#pragma unroll
          for (int q = 0; q < i; ++q) acc = __builtin_fma(-ev.Lm[i][q], U[t][q], acc);
          U[t][i] = acc * ev.invd[i];
          nq = __builtin_fma(U[t][i], U[t][i], nq);
        }
        const double var = ev.sigma_sq * nq;
        if (var > LMM_NEAR_ZERO && var < __builtin_inf()) tsq[t] = est * est / var;
      }
    }
    // joint: b_T' Cov_TT^-1 b_T with Cov_ab = sigma^2 U_a.U_b (Cholesky of the nt x nt block)
    bool joint_sig = false;
    if (conv && tkeep != 0u) {
      double Lc[MAX_T_WAVE][MAX_T_WAVE], cj[MAX_T_WAVE];
      bool jok = true;
      double qsum = 0.0;
#pragma unroll
      for (int j = 0; j < MAX_T_WAVE; ++j) {
        if ((uint32_t)j < nt && !((tkeep >> j) & 1u)) {  // untestable target: identity row, no contribution
          Lc[j][j] = 1.0;
#pragma unroll
          for (int i2 = j + 1; i2 < MAX_T_WAVE; ++i2) Lc[i2][j] = 0.0;
        }
        if ((tkeep >> j) & 1u) {
          double dd = 0.0;
#pragma unroll
          for (int i = 0; i < P; ++i) dd = __builtin_fma(U[j][i], U[j][i], dd);
          dd *= ev.sigma_sq;
#pragma unroll
          for (int t = 0; t < j; ++t) dd = __builtin_fma(-Lc[j][t], Lc[j][t], dd);
          jok = jok && (dd > 0.0);
          Lc[j][j] = __builtin_sqrt(dd);
#pragma unroll
          for (int i2 = j + 1; i2 < MAX_T_WAVE; ++i2) {
            if ((uint32_t)i2 < nt) {
              double vv = 0.0;  // U of an untestable target is all zeros
#pragma unroll
              for (int i = 0; i < P; ++i) vv = __builtin_fma(U[i2][i], U[j][i], vv);
              vv *= ev.sigma_sq;
#pragma unroll
              for (int t = 0; t < j; ++t) vv = __builtin_fma(-Lc[i2][t], Lc[j][t], vv);
              Lc[i2][j] = vv / Lc[j][j];
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MAX_T_WAVE; ++i) {
        if ((uint32_t)i < nt) {
          const double* c = prm.contrast + i * MAX_P;
          double vv = 0.0;
#pragma unroll
          for (int q = 0; q < P; ++q) vv = __builtin_fma(((tkeep >> i) & 1u) ? c[q] : 0.0, ev.beta[q], vv);
#pragma unroll
          for (int t = 0; t < i; ++t) vv = __builtin_fma(-Lc[i][t], cj[t], vv);
          cj[i] = vv / Lc[i][i];
          qsum = __builtin_fma(cj[i], cj[i], qsum);
        }
      }
      // the full table row holds chi2(n_targets); a reduced round tests its k_red kept targets (batch.rs:1456-1468)
      const double joint_crit = reduced_round ? prm.joint_crit_k[__popc(tkeep)] : crow[2 + nt];
      joint_sig = jok && (qsum < __builtin_inf()) && (qsum > joint_crit);
    }
    uint32_t unc_bits = 0, cor_bits = 0;
    decide_tests(tsq, nt, crow, prm.correction, unc_bits, cor_bits);
    if (!conv) {
      unc_bits = 0;
      cor_bits = 0;
    }
    if (active) {
      const size_t cell = (size_t)out_row * prm.n_grid + g;
#pragma unroll
      for (int t = 0; t < MAX_T_WAVE; ++t)
        if ((uint32_t)t < nt) {
          prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
          prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
        }
      prm.conv[cell] = conv ? 1 : 0;
      prm.boundary[cell] = (uint8_t)bh;
      prm.joint[cell] = joint_sig ? 1 : 0;
      if (RED && prm.factor_excluded)
        for (uint32_t f = 0; f < prm.n_factors; ++f)
          prm.factor_excluded[cell * prm.n_factors + f] = (uint8_t)((flag_bits >> f) & 1u);
      if (prm.overall) prm.overall[cell] = 0;  // batch.rs:1449-1451
      if (prm.stat && g == 0) {
#pragma unroll
        for (int t = 0; t < MAX_T_WAVE; ++t)
          if ((uint32_t)t < nt) prm.stat[(size_t)out_row * nt + t] = tsq[t];
      }
    }
  }
}

size_t lmm_lds_bytes(int P, uint32_t max_n, uint32_t ncl_max) {
  const size_t n_pad = (max_n + 3u) & ~3u;
  const size_t ncl_pad = (ncl_max + 1u) & ~(size_t)1u;
  return ((size_t)P * n_pad + ncl_pad + (size_t)4 * (P + 1) * LWAVE) * sizeof(float);
}

// doubles of record storage per (replicate, grid point)
size_t lmm_rec_fields(int P, uint32_t) { return (size_t)(P + 1) * (P + 2) + 4 + (size_t)P * (P + 1) / 2 + P + 1; }

template <bool FACTORS, bool RED, bool SCEN>
static hipError_t launch_lmm_prepare(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max,
                                     hipStream_t st) {
  const size_t lds = lmm_lds_bytes(P, max_n, ncl_max);
  const dim3 block(LWAVE), grid(n_sims);
#define LL(PP)                                                                                                        \
  case PP: {                                                                                                          \
    if (lds > 48 * 1024) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)lmm_prepare_kernel<PP, FACTORS, RED, SCEN>,                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return e;                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((lmm_prepare_kernel<PP, FACTORS, RED, SCEN>), grid, block, lds, st, d_prm);                    \
  } break;
  switch (P) {
    LL(2) LL(3) LL(4) LL(5) LL(6) LL(7) LL(8)
    default: return hipErrorInvalidValue;
  }
#undef LL
  return hipGetLastError();
}

template <bool RED>
static hipError_t launch_lmm_optimise(int P, const SimParams* d_prm, uint32_t n_sims, hipStream_t st) {
  const dim3 block(64), grid((n_sims + 63u) / 64u);
  switch (P) {
    case 2: hipLaunchKernelGGL((lmm_optimise_kernel<2, RED>), grid, block, 0, st, d_prm); break;
    case 3: hipLaunchKernelGGL((lmm_optimise_kernel<3, RED>), grid, block, 0, st, d_prm); break;
    case 4: hipLaunchKernelGGL((lmm_optimise_kernel<4, RED>), grid, block, 0, st, d_prm); break;
    case 5: hipLaunchKernelGGL((lmm_optimise_kernel<5, RED>), grid, block, 0, st, d_prm); break;
    case 6: hipLaunchKernelGGL((lmm_optimise_kernel<6, RED>), grid, block, 0, st, d_prm); break;
    case 7: hipLaunchKernelGGL((lmm_optimise_kernel<7, RED>), grid, block, 0, st, d_prm); break;
    case 8: hipLaunchKernelGGL((lmm_optimise_kernel<8, RED>), grid, block, 0, st, d_prm); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_lmm_sim(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max, bool factors,
                          bool reduced, bool scen, hipStream_t st) {
  hipError_t e;
  if (scen) e = launch_lmm_prepare<true, true, true>(P, d_prm, n_sims, max_n, ncl_max, st);
  else if (reduced) e = launch_lmm_prepare<true, true, false>(P, d_prm, n_sims, max_n, ncl_max, st);
  else if (factors) e = launch_lmm_prepare<true, false, false>(P, d_prm, n_sims, max_n, ncl_max, st);
  else e = launch_lmm_prepare<false, false, false>(P, d_prm, n_sims, max_n, ncl_max, st);
  if (e != hipSuccess) return e;
  return (scen || reduced) ? launch_lmm_optimise<true>(P, d_prm, n_sims, st) : launch_lmm_optimise<false>(P, d_prm, n_sims, st);
}

}  // namespace mcp
