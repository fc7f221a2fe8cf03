// mcp_rows.hpp -- per-replicate data generation shared by the OLS (lane-per-replicate) and GLM / LMM
// (wave-per-replicate) kernels: 4-row Philox blocks staged in LDS with the deferred inverse-CDF tail
// fix-up, and the row builder (L.z mix, factor dummies, interactions, outcome).
// Reference: crates/engine-core/src/data_gen.rs:87-925 (line ranges at each function).
#pragma once
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_scenario.hpp"

#pragma clang fp contract(off)

namespace mcp {

// Philox key of a replicate: mix13(call_seed ^ rotl64(sim_id, 32)) (rng.rs:36-37, 61-69).  Under
// MCP_SEED_PER_REPLICATE (validation aid of the clustered-GLM kernels) replicate i is replicate 0 of the call seed
// splitmix64(base_seed + i): the reference's per-draw harness (validation/oracle2_power.R:38-66)
__device__ __forceinline__ uint64_t replicate_key(const SimParams& prm, uint64_t sim_id) {
  return prm.seed_per_replicate ? pcg_mix64(splitmix64(prm.base_seed + sim_id), 0ull) : pcg_mix64(prm.base_seed, sim_id);
}

// categorical draw on the sequential stream (rng.rs:112-127)
__device__ __forceinline__ uint32_t categorical(const SimParams& prm, uint32_t f, uint32_t k0, uint32_t k1, uint64_t q) {
  const double u = (double)u32_to_unit_f32(sequential_word(k0, k1, q));
  const uint32_t nl = prm.f_levels[f];
  const uint32_t off = prm.f_prop_off[f];
  double acc = 0.0;
  uint32_t lvl = nl - 1;
  bool found = false;
  for (uint32_t i = 0; i < nl; ++i) {
    acc += prm.f_props[off + i];
    if (!found && u < acc) {
      lvl = i;
      found = true;
    }
  }
  return lvl;
}

// ---- 4-row block staging ------------------------------------------------------------------
// LDS layout: stage[slot * 256 + tid], slot = 4 * column + r, column j < K = continuous predictor j,
// column K = the residual.  One Philox call yields rows 4*ctr..4*ctr+3 of one column (rng.rs:149-178).

// pending-tail mask: 4 bits per staged column; more than 8 columns need 64 bits
template <int P>
struct PendMask { typedef uint32_t type; };
template <> struct PendMask<9> { typedef unsigned long long type; };
template <> struct PendMask<10> { typedef unsigned long long type; };
template <> struct PendMask<11> { typedef unsigned long long type; };
template <> struct PendMask<12> { typedef unsigned long long type; };
template <> struct PendMask<13> { typedef unsigned long long type; };
template <> struct PendMask<14> { typedef unsigned long long type; };
template <> struct PendMask<15> { typedef unsigned long long type; };
template <> struct PendMask<16> { typedef unsigned long long type; };  // 16 slots x 4 rows: the last size one u64 holds
__device__ __forceinline__ uint32_t pend_ctz(uint32_t m) { return (uint32_t)__builtin_ctz(m); }
__device__ __forceinline__ uint32_t pend_ctz(unsigned long long m) { return (uint32_t)__builtin_ctzll(m); }

template <int STRIDE, typename MaskT>
__device__ __forceinline__ void stage_column(float* st, uint32_t col_slot, const u32x4 w, MaskT& pend) {
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = u32_to_unit_f32(ws[r]) - 0.5f;
    const bool tail = __builtin_fabsf(v) > NORM_INV_BP;
    // branch-free central polynomial for every lane.  (Forcing v_pk_mul_f32 / v_pk_add_f32 over draw pairs was
    // measured 8 % SLOWER on the fused kernel: packed f32 issues at ~1.6x the scalar cost here, profiles/README.)
    const float z = icdf_central(v);
    st[(col_slot * 4u + r) * STRIDE] = tail ? v : z;  // tail lanes park v for the fix-up
    pend |= tail ? ((MaskT)1 << (col_slot * 4u + r)) : (MaskT)0;
  }
}

// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside each row of 16 with DPP row shifts,
// then the row totals carried across with row_bcast15 / row_bcast31 (gfx9 DPP)
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast15 into rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast31 into rows 2, 3
  return x;
}

constexpr uint32_t TAIL_QUEUE = 128;  // entries per wave of the dense tail queue

// Tail fix-up of one staged block.  ~4.85 % of the draws need the tail branch of the inverse CDF; a per-lane loop
// runs max_lane(count) passes of the ~75-instruction branch with most lanes idle (3-4 passes for ~0.8 parked draws
// per lane).  The dense form compacts the wave's parked draws into an LDS queue (prefix sum over the lanes) and
// lets all 64 lanes work them off: ~50 draws per wave and block = one full pass.
template <int STRIDE, typename MaskT>
__device__ __forceinline__ void fixup_tails(float* st, MaskT pend, uint32_t* queue) {
  if (queue != nullptr) {
    const uint32_t cnt = (uint32_t)__popcll((unsigned long long)pend);
    const uint32_t incl = wave_inclusive_scan_u32(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0u) return;
    if (total <= TAIL_QUEUE) {
      uint32_t off = incl - cnt;
      MaskT m = pend;
      while (m != 0) {  // each parked draw: its index in the block's staging area
        const uint32_t slot = pend_ctz(m);
        m &= m - 1;
        queue[off++] = slot * STRIDE + threadIdx.x;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float* base = st - threadIdx.x;
      for (uint32_t j = threadIdx.x & 63u; j < total; j += 64u) {
        const uint32_t idx = queue[j];
        base[idx] = icdf_tail(base[idx]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      return;
    }
  }
  while (__ballot(pend != 0) != 0ull) {
    if (pend != 0) {
      const uint32_t slot = pend_ctz(pend);
      pend &= pend - 1;
      st[slot * STRIDE] = icdf_tail(st[slot * STRIDE]);
    }
  }
}

// rows 4*ctr .. 4*ctr+3 of one replicate: standard normals of every column into LDS (data_gen.rs:316-421)
// OUTCOME: 0 = continuous (normal residual), 1 = discrete GLM outcome by prm.glm_family: logit keeps the
// Uniform(0,1) draw in the residual slot, probit and Poisson a standard normal (data_gen.rs:390-411)
template <int P, int STRIDE, int OUTCOME, bool SCEN = false>
__device__ __forceinline__ void stage_group(const SimParams& prm, uint32_t k0, uint32_t k1, uint32_t ctr, float* st,
                                            uint32_t scen_meta = 0u, uint32_t* tail_queue = nullptr) {
  const uint32_t K = prm.K;
  typedef typename PendMask<P>::type mask_t;
  mask_t pend = 0;
  float eu[4] = {0.f, 0.f, 0.f, 0.f};  // slot-0 uniforms (Uniform residual, data_gen.rs:485-501)
#pragma unroll
  for (int j = 0; j < P - 1; ++j)
    if ((uint32_t)j < K) stage_column<STRIDE>(st, (uint32_t)j, planar_words(k0, k1, CLASS_XNORM, (uint32_t)j, ctr), pend);
  if (OUTCOME == 0 || prm.glm_family != 0u) {
    const u32x4 w0 = planar_words(k0, k1, CLASS_RESID, 0u, ctr);
    stage_column<STRIDE>(st, K, w0, pend);
    if (SCEN) {
      eu[0] = u32_to_unit_f32(w0.x);
      eu[1] = u32_to_unit_f32(w0.y);
      eu[2] = u32_to_unit_f32(w0.z);
      eu[3] = u32_to_unit_f32(w0.w);
    }
  } else {
    const u32x4 w = planar_words(k0, k1, CLASS_RESID, 0u, ctr);
    st[(K * 4u + 0u) * STRIDE] = u32_to_unit_f32(w.x);
    st[(K * 4u + 1u) * STRIDE] = u32_to_unit_f32(w.y);
    st[(K * 4u + 2u) * STRIDE] = u32_to_unit_f32(w.z);
    st[(K * 4u + 3u) * STRIDE] = u32_to_unit_f32(w.w);
  }
  fixup_tails<STRIDE, mask_t>(st, pend, tail_queue);
  if (SCEN && OUTCOME == 0) {
    // residual shapes (data_gen.rs:412-501): slot 0 = z; t / chi-square forms add the normals of slots 1..df.
    // The realized distribution is per replicate (scenario swap), the column count is launch-uniform.
    const uint32_t rd = (scen_meta >> 28) & 7u;
    const double df = prm.resid_df;
    double chi_hk[4] = {0.0, 0.0, 0.0, 0.0}, chi_sk[4] = {0.0, 0.0, 0.0, 0.0};
    float z0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z0[r] = st[(K * 4u + r) * STRIDE];
    // the extra columns feed the chi-square forms only: a wave none of whose lanes holds one skips their draws (Philox
    // is counter-based, nothing downstream depends on the skipped words)
    if (prm.resid_extra_cols && __builtin_amdgcn_ballot_w64(rd >= 1u && rd <= 3u) != 0ull) {
      const uint32_t dfi = prm.resid_extra_cols - 1u;
#pragma unroll
      for (int r = 0; r < 4; ++r) chi_sk[r] = (double)z0[r] * (double)z0[r];  // skewed forms start at slot 0
      for (uint32_t s = 1; s <= dfi; ++s) {
        unsigned long long pd = 0;  // slot K + 1 is past bit 31 of the narrow mask once K >= 7
        stage_column<STRIDE>(st, K + 1u, planar_words(k0, k1, CLASS_RESID, s, ctr), pd);
        fixup_tails<STRIDE, unsigned long long>(st, pd, nullptr);  // ~12 parked draws per wave: the loop is cheaper than a scan
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double g = (double)st[((K + 1u) * 4u + r) * STRIDE];
          chi_hk[r] = chi_hk[r] + g * g;             // HighKurtosis: slots 1..df
          if (s < dfi) chi_sk[r] = chi_sk[r] + g * g;  // Right/LeftSkewed: slots 0..df-1
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float e = z0[r];
      if (rd == 1u) {  // HighKurtosis: t(df) standardized
        const double z = (double)z0[r];
        const double denom = __builtin_sqrt(chi_hk[r] / df);
        const double t = denom <= 0.0 ? z : z / denom;
        e = (float)(t * (1.0 / __builtin_sqrt(df / (df - 2.0))));
      } else if (rd == 2u || rd == 3u) {  // Right / LeftSkewed: centered scaled chi-square
        const double sign = rd == 3u ? -1.0 : 1.0;
        e = (float)(sign * (chi_sk[r] - df) * (1.0 / __builtin_sqrt(2.0 * df)));
      } else if (rd == 4u) {  // Uniform(-sqrt3, sqrt3)
        e = (float)((2.0 * (double)eu[r] - 1.0) * SQRT3);
      }
      st[(K * 4u + r) * STRIDE] = e;
    }
  }
}

// per-replicate design record of the general ("scenario") path, loaded once per lane
template <int P>
struct ScenLane {
  double Lp[(P - 1) * P / 2 > 0 ? (P - 1) * P / 2 : 1];  // row-major lower factor: idx(j, t) = j(j+1)/2 + t
  double beta_eff[P];
  uint32_t meta;  // 3 bits per predictor marginal (K <= 9) | residual dist << 28 | not-PSD flag << 31
};

template <int P>
__device__ __forceinline__ void scen_lane_load(const SimParams& prm, uint32_t lsim, ScenLane<P>& sl) {
  const uint32_t K = prm.K;
  sl.meta = prm.scen_meta[lsim];
#pragma unroll
  for (int i = 0; i < (P - 1) * P / 2; ++i)
    sl.Lp[i] = ((uint32_t)i < K * (K + 1u) / 2u) ? prm.scen_L[(size_t)i * prm.scen_stride + lsim] : 0.0;
#pragma unroll
  for (int c = 0; c < P; ++c)
    sl.beta_eff[c] = prm.scen_beta ? prm.scen_beta[(size_t)c * prm.scen_stride + lsim] : prm.beta[c];
}

// launch-constant values of the row loop, hoisted into scalar registers once per wave
template <int P>
struct RowCtx {
  uint32_t K;
  uint32_t ident_L;
  uint32_t n_factors;        // 0 on the continuous-only fast path (FACTORS == false)
  uint32_t beta_nz;          // bit c set when beta[c] != 0
  uint32_t inter_any;        // bit c set when column c is an interaction
  uint32_t imask[P];
  double beta[P];
};

template <int P, bool FACTORS>
__device__ __forceinline__ void row_ctx_init(const SimParams& prm, RowCtx<P>& cx) {
  cx.K = prm.K;
  cx.ident_L = prm.ident_L;
  cx.n_factors = FACTORS ? prm.n_factors : 0u;
  cx.beta_nz = 0u;
  cx.inter_any = 0u;
#pragma unroll
  for (int c = 0; c < P; ++c) {
    cx.beta[c] = prm.beta[c];
    cx.imask[c] = prm.inter_mask[c];
    cx.beta_nz |= (cx.beta[c] != 0.0) ? (1u << c) : 0u;
    cx.inter_any |= (cx.imask[c] != 0u) ? (1u << c) : 0u;
  }
}

// the design row: continuous block, factor dummies, bootstrap copies, interactions; returns the residual draw
// PREMIXED (general path, lane-per-replicate kernel): the staging area already holds the mixed and transformed values
// of the continuous columns (scen_premix_quad below)
template <int P, bool FACTORS, int STRIDE, bool SCEN, bool PREMIXED = false>
__device__ __forceinline__ float build_x(const SimParams& prm, const RowCtx<P>& cx, uint32_t k0, uint32_t k1,
                                         const float* st, int r, uint32_t row, float (&x)[P], const ScenLane<P>* sl) {
  const uint32_t K = cx.K;
  float z[P - 1];
#pragma unroll
  for (int j = 0; j < P - 1; ++j) z[j] = ((uint32_t)j < K) ? st[(j * 4 + r) * STRIDE] : 0.0f;
  const float e = st[(K * 4u + r) * STRIDE];
  x[0] = 1.0f;
  // factor levels of this row (data_gen.rs:513-551)
  uint32_t lvl[MAX_FACTORS];
  if (FACTORS) {
#pragma unroll
    for (int f = 0; f < MAX_FACTORS; ++f) {
      lvl[f] = 0u;
      if ((uint32_t)f < cx.n_factors) {
        if (prm.f_sampled[f]) {
          const uint64_t q = (uint64_t)prm.n_seq_pre + (uint64_t)row * prm.n_seq_per_row + prm.f_seq_slot[f];
          lvl[f] = categorical(prm, (uint32_t)f, k0, k1, q);
        } else {
          lvl[f] = row < prm.max_n ? (uint32_t)prm.fixed_levels[(size_t)row * prm.n_factors + f] : 0u;
        }
      }
    }
  }
  uint32_t boot_row = 0;
  if (SCEN && prm.bootstrap) {  // one source-row index per design row, drawn after the categorical draws (data_gen.rs:562-566)
    const uint64_t q = (uint64_t)prm.n_seq_pre + (uint64_t)row * prm.n_seq_per_row + prm.n_sampled_factors;
    const double u = (double)u32_to_unit_f32(sequential_word(k0, k1, q));
    boot_row = (uint32_t)(u * (double)prm.upload_data_rows);
    boot_row = boot_row > prm.upload_data_rows - 1u ? prm.upload_data_rows - 1u : boot_row;
    if (FACTORS) {  // uploaded factors re-expand the copied level code (data_gen.rs:590-603)
#pragma unroll
      for (int f = 0; f < MAX_FACTORS; ++f) {
        if ((uint32_t)f < cx.n_factors && prm.boot_map[K + f] >= 0) {
          const double lv = floor(prm.upload_data[(size_t)boot_row * prm.upload_data_cols + (uint32_t)prm.boot_map[K + f]] + 0.5);
          const uint32_t level = lv > 0.0 ? (uint32_t)lv : 0u;
          lvl[f] = (level > 0u && level - 1u < prm.f_levels[f] - 1u) ? level : 0u;
        }
      }
    }
  }
#pragma unroll
  for (int c = 1; c < P; ++c) {
    float v = 0.0f;
    if ((uint32_t)(c - 1) < K) {
      if (SCEN) {  // per-replicate factor and marginal (data_gen.rs:334-363)
        if (PREMIXED) {
          v = z[c - 1];
        } else {
          double acc = 0.0;
#pragma unroll
          for (int t = 0; t < c; ++t) acc = acc + sl->Lp[(c - 1) * c / 2 + t] * (double)z[t];
          v = (float)acc;
          const uint32_t vt = (sl->meta >> (3 * (c - 1))) & 7u;
          if (vt != 0u)
            v = (float)apply_marginal(vt, (double)v, prm.var_params[c - 1], prm.t3_table, prm.upload_normal,
                                      prm.upload_rows, K, (uint32_t)(c - 1));
        }
        if (prm.bootstrap && prm.boot_map[c - 1] >= 0) {  // strict mode copies the source row's value (data_gen.rs:566-589)
          double bv = prm.upload_data[(size_t)boot_row * prm.upload_data_cols + (uint32_t)prm.boot_map[c - 1]];
          if (prm.var_types[c - 1] == 7u) bv = (bv + prm.var_params[c - 1] >= 0.5) ? 1.0 : 0.0;  // stored = x - p
          v = (float)bv;
        }
      } else if (cx.ident_L) {
        v = z[c - 1];
      } else {  // x_j = f32( sum_{t<=j} L[j][t] * f64(z_t) ), t ascending, separately rounded (data_gen.rs:334-351)
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < c; ++t) {
          const double l = prm.L[(c - 1) * MAX_K + t];
          if (l != 0.0) acc = acc + l * (double)z[t];
        }
        v = (float)acc;
      }
    } else if ((cx.inter_any >> c) & 1u) {  // raw product in f64, components ascending (data_gen.rs:643-649)
      double prod = 1.0;                    // non-members multiply by exactly 1.0: no branch, same bits
#pragma unroll
      for (int cc = 1; cc < c; ++cc) prod = prod * (((cx.imask[c] >> cc) & 1u) ? (double)x[cc] : 1.0);
      v = (float)prod;
    } else if (FACTORS && prm.col_factor[c] != 0xffffffffu) {
      const uint32_t f = prm.col_factor[c];
      const uint32_t lv = f == 0 ? lvl[0] : (f == 1 ? lvl[1] : (f == 2 ? lvl[2] : lvl[3]));
      v = (lv == prm.col_level[c]) ? 1.0f : 0.0f;
    }
    x[c] = v;
  }
  return e;
}

// General path, lane-per-replicate kernel: the per-replicate correlated mix x_c = f32(sum_{t < c} L[c-1][t] f64(z_t)) and the
// marginal transform of every continuous column, for the four staged rows, in place: slot (c - 1, r) of the staging area
// ends up holding what build_x<.., SCEN> computes for x[c] (same operations, same order; data_gen.rs:334-363).  Columns
// descend, so the normals a column still needs are intact when its slot is overwritten.  The transform runs as ONE loop
// over the 4 K slots: a single copy of apply_marginal in the instruction stream.
template <int P, int STRIDE>
__device__ __forceinline__ void scen_premix_quad(const SimParams& prm, float* st, const ScenLane<P>& sl) {
  const uint32_t K = prm.K;
  // (the factor stays in the lane's registers: re-read from the replicate's record in HBM at every quad it frees 30 VGPRs
  // and costs 15 % of the kernel -- two waves per SIMD do not hide the loads)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float z[P - 1];
#pragma unroll
    for (int j = 0; j < P - 1; ++j) z[j] = ((uint32_t)j < K) ? st[(j * 4 + r) * STRIDE] : 0.0f;
#pragma unroll
    for (int c = P - 1; c >= 1; --c) {
      if ((uint32_t)(c - 1) < K) {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < c; ++t) acc = acc + sl.Lp[(c - 1) * c / 2 + t] * (double)z[t];
        st[((c - 1) * 4 + r) * STRIDE] = (float)acc;
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64((sl.meta & 0x0fffffffu) != 0u) == 0ull) return;  // no lane of the wave holds a swapped marginal
#pragma nounroll
  for (uint32_t slot = 0; slot < 4u * K; ++slot) {
    const uint32_t j = slot >> 2;
    const uint32_t vt = (sl.meta >> (3u * j)) & 7u;
    if (vt != 0u) {
      const float v = st[slot * STRIDE];
      st[slot * STRIDE] = (float)apply_marginal(vt, (double)v, prm.var_params[j], prm.t3_table, prm.upload_normal, prm.upload_rows, K, j);
    }
  }
}

// interaction columns recomputed from (possibly overwritten) component columns (data_gen.rs:643-649)
template <int P>
__device__ __forceinline__ void rebuild_interactions(const RowCtx<P>& cx, float (&x)[P]) {
#pragma unroll
  for (int c = 1; c < P; ++c) {
    if ((cx.inter_any >> c) & 1u) {
      double prod = 1.0;
#pragma unroll
      for (int cc = 1; cc < c; ++cc) prod = prod * (((cx.imask[c] >> cc) & 1u) ? (double)x[cc] : 1.0);
      x[c] = (float)prod;
    }
  }
}

// the outcome of a finished design row (data_gen.rs:781-922)
template <int P, int OUTCOME, bool SCEN>
__device__ __forceinline__ void finish_row(const SimParams& prm, const RowCtx<P>& cx, const float (&x)[P], float e,
                                           float& y, double u_re, const ScenLane<P>* sl) {
  // y = f32( sum_j f64(x_ij) beta_j + f64(eps_i) + u_re )  (data_gen.rs:785-857); zero betas add exact zeros
  double lp = 0.0;
#pragma unroll
  for (int c = 0; c < P; ++c)
    if ((cx.beta_nz >> c) & 1u) lp = lp + (double)x[c] * cx.beta[c];
  if (SCEN) {
    const double lp_clean = lp;
    if (prm.het_total > SCEN_NEAR_ZERO) {  // per-study effects (data_gen.rs:727-751, 785-796)
      double lpe = 0.0;
#pragma unroll
      for (int c = 0; c < P; ++c) lpe = lpe + (double)x[c] * sl->beta_eff[c];
      lp = lpe;
    }
    if (OUTCOME == 0 && prm.apply_hsk) {  // Var(eps_i) = exp(gamma z_i) / exp(gamma^2 / 2) (data_gen.rs:798-809, 876-888)
      double drv = lp_clean;
      if (prm.hsk_driver >= 0) {
#pragma unroll
        for (int c = 0; c < P; ++c) drv = (c == prm.hsk_driver) ? (double)x[c] : drv;
      }
      const double arg = prm.hsk_gamma * ((drv - prm.hsk_center) * prm.hsk_inv_std);
      const double mult = exp(arg) * prm.hsk_inv_norm;
      const float es = (float)((double)e * __builtin_sqrt(mult));
      y = (float)((lp + u_re) + (double)es);
      return;
    }
  }
  if (OUTCOME == 0) {
    y = (float)((lp + (double)e) + u_re);  // u_re = 0.0 without a cluster (data_gen.rs:814-857)
  } else {
    const double eta = lp + u_re;
    if (prm.glm_family == 0u) {  // Bernoulli(sigmoid(eta)): 1{u < sigma(eta)} with the pre-drawn uniform (data_gen.rs:890-911)
      const double pr = eta >= 0.0 ? 1.0 / (1.0 + exp(-eta)) : exp(eta) / (1.0 + exp(eta));
      y = ((double)e < pr) ? 1.0f : 0.0f;
    } else if (prm.glm_family == 1u) {  // probit: eta is the latent threshold of the N(0,1) draw (data_gen.rs:899-907)
      y = ((double)e < eta) ? 1.0f : 0.0f;
    } else {  // Poisson by fixed-draw inverse CDF (rng.rs:370-419): exact walk for lambda <= 20, normal rounding above
      const double lam = exp(eta), z = (double)e;
      double out = 0.0;
      if (lam > 0.0) {
        if (lam <= 20.0) {
          const double u = phi_as(z);
          double pk = exp(-lam), cdf = pk;
          uint32_t kk = 0;
          while (cdf < u && kk < 200u) {
            kk += 1u;
            pk *= lam / (double)kk;
            cdf += pk;
          }
          out = (double)kk;
        } else {
          const double v = floor(lam + z * __builtin_sqrt(lam) + 0.5);
          out = v > 0.0 ? v : 0.0;
        }
      }
      y = (float)out;
    }
  }
}

// one design row + outcome from the staged normals (data_gen.rs:334-363, 509-551, 643-649, 781-857)
template <int P, bool FACTORS, int STRIDE, int OUTCOME, bool SCEN = false>
__device__ __forceinline__ void build_row(const SimParams& prm, const RowCtx<P>& cx, uint32_t k0, uint32_t k1,
                                          const float* st, int r, uint32_t row, float (&x)[P], float& y,
                                          double u_re = 0.0, const ScenLane<P>* sl = nullptr) {
  const float e = build_x<P, FACTORS, STRIDE, SCEN>(prm, cx, k0, k1, st, r, row, x, sl);
  finish_row<P, OUTCOME, SCEN>(prm, cx, x, e, y, u_re, sl);
}

}  // namespace mcp
