// ols_wide_kernels.hip -- OLS fits of WIDE designs (13 .. 16 design columns; the reference bench's `ols_wide` =
// base_spec(15), P = 16: crates/engine-core/src/bin/throughput.rs:274-277), one sample size per call.
//
// The lane-per-replicate kernels (ols_kernels.hip) keep a replicate's sufficient statistics in ONE lane's registers:
// the lower triangle of [X | y]'[X | y] is (P + 1)(P + 2) / 2 doubles -- 153 at P = 16 = 306 VGPRs.  Past 256
// architectural registers the compiler parks accumulators in AGPRs (every multiply-add then costs a v_accvgpr_read
// and a v_accvgpr_write beside it) and spills the rest: the run-time-sized reduced-design kernel that served these
// widths carried 1.2-2.5 KB of scratch per lane, 6.5 GB of spill traffic per 10^6-fit launch, 5.7e7 fits/s
// (profiles/r01_i_ols_wide_pmc.csv).  MFMA does not help (f64 MFMA peak = f64 VALU peak,
// operands differ per replicate).
//
// Here FOUR WAVES share 64 replicates (workgroup = 256 threads, lane l of every wave = replicate l):
//   * generation is split by draw column: wave w stages the Philox / inverse-CDF blocks of the columns j = w (mod 4)
//     into LDS ([slot][lane]); after a barrier wave w builds row w of the 4-row block with the bit-exact row builder of
//     mcp_rows.hpp and leaves it in LDS widened to f64 (a row is built and converted once, not once per wave); after a
//     second barrier every wave reads the four rows;
//   * accumulation is split by ROW of the triangle: role k owns a band of rows holding about a quarter of the entries
//     (P = 16: rows 0-8 | 9-11 | 12-14 | 15-16 = 45, 33, 42, 33 doubles): every accumulator is an architectural VGPR, the
//     role test is wave-uniform (no divergence, no cross-lane traffic), 0 B of scratch;
//   * the fit runs in place down the chain of roles: a role completes its rows against the finished rows of the roles
//     before it (published through LDS, one chunk each), factors its own pivots and publishes; the solve runs back up the
//     chain, the test statistics down again, the last role (it owns the outcome row) decides.  Per entry the factor is
//     formed by the operation sequence of ols_kernels.hip's fit_decide (same fma chains, t ascending): L and every
//     statistic whose chain stays inside one role are bit-identical to it; the sums that cross roles (beta, bxty) differ
//     in summation order only (t^2 within 1e-13 of the oracle).
// Semantics: batch.rs:871-1142 (OLS arm), posthoc.rs:86-99 (contrast statistic), correction.rs:16-92.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"
#include "mcp_suff.hpp"
#include "mcp_ols_finish.hpp"

#pragma clang fp contract(off)

namespace mcp {
namespace {

constexpr int WR = 4;        // waves (roles) sharing the 64 replicates of a workgroup
constexpr int WB = 64 * WR;  // threads per workgroup
constexpr int WL = 64;       // replicates per workgroup = lanes per wave

constexpr int tri(int i) { return i * (i + 1) / 2; }

// fix-up of the parked tail draws of THIS wave's staged columns (mcp_rows.hpp fixup_tails, with the staging column of a
// lane given explicitly: both waves of the workgroup stage for replicate = lane)
template <typename MaskT>
__device__ __forceinline__ void wide_fixup_tails(float* buf, uint32_t lane, MaskT pend, uint32_t* queue) {
  const uint32_t cnt = (uint32_t)__popcll((unsigned long long)pend);
  const uint32_t incl = wave_inclusive_scan_u32(cnt);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (total == 0u) return;
  if (total <= TAIL_QUEUE) {
    uint32_t off = incl - cnt;
    MaskT m = pend;
    while (m != 0) {
      const uint32_t slot = pend_ctz(m);
      m &= m - 1;
      queue[off++] = slot * WL + lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < total; j += 64u) {
      const uint32_t idx = queue[j];
      buf[idx] = icdf_tail(buf[idx]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return;
  }
  float* st = buf + lane;
  while (__ballot(pend != 0) != 0ull) {
    if (pend != 0) {
      const uint32_t slot = pend_ctz(pend);
      pend &= pend - 1;
      st[slot * WL] = icdf_tail(st[slot * WL]);
    }
  }
}

// rows [wide_lo(V, k), wide_lo(V, k + 1)) of the V x V lower triangle belong to role k: the smallest boundary that gives the
// roles before it k / WR of the entries (V = 17: 0 | 9 | 12 | 15 | 17 -> 45, 33, 42, 33 entries)
constexpr int wide_lo(int V, int k) {
  if (k <= 0) return 0;
  if (k >= WR) return V;
  int b = 0;
  while (tri(b) * WR < k * tri(V)) ++b;
  return b;
}

// One role's share of the work on a finished row and of the fit.  acc holds the role's rows packed by rows:
// entry (i, j <= i) at acc[tri(i) + j - tri(LO)].
template <int P, int ROLE>
struct WideRole {
  static constexpr int V = P + 1;
  static constexpr int LO = wide_lo(V, ROLE), HI = wide_lo(V, ROLE + 1);
  static constexpr int NE = tri(HI) - tri(LO);
  static constexpr bool LAST = ROLE == WR - 1;
  static constexpr int HIP = HI < P ? HI : P;  // rows that are design columns (the last role also owns the outcome row P)

  template <int NACC>
  static __device__ __forceinline__ void add_row(double (&acc)[NACC], const double (&v)[P + 1]) {
    // entry (i, j <= i) of [1 | X | y]'[1 | X | y]: sums against the intercept are plain adds, the rest fma -- the
    // arithmetic of suff_add_row (mcp_suff.hpp); (0, 0) = n is never accumulated
#pragma unroll
    for (int i = LO; i < HI; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        if (i == 0) continue;
        double& a = acc[tri(i) + j - tri(LO)];
        a = j == 0 ? a + v[i] : __builtin_fma(v[i], v[j], a);
      }
  }
};

// Exchange area of the fit (the staging buffers, dead by then): xb[k * WL], 64 doubles per lane.
//   factor phase : a role's rows packed by rows, its reciprocal diagonals, then (ok, dmin, dmax) so far
//   solve phase  : beta_i at [i], the running right-hand side w_i at [P + i]
//   test phase   : u of two tests at [tt * P + i], the squared statistics at [2 * P + t]
template <int P, int ROLE, int NACC>
__device__ __forceinline__ void wide_fit(const SimParams& prm, double (&acc)[NACC], double* xb, uint32_t n, uint32_t g, bool active, uint32_t out_row) {
  typedef WideRole<P, ROLE> RL;
  constexpr int LO = RL::LO, HI = RL::HI, HIP = RL::HIP, V = P + 1;
  constexpr int MT = MAX_T;
  constexpr int NOWN = HIP - LO;  // pivots owned
#define LR(i, j) acc[tri(i) + (j) - tri(LO)]
  static_assert(RL::NE <= NACC, "accumulator array too small");
  static_assert(RL::LAST || (RL::NE + (HI - LO) + 3 <= 64), "a role's rows must fit one exchange chunk");
  static_assert(2 * P + MT <= 64, "test exchange");
  double rhs[RL::LAST ? P : 1];  // X'y before the factorisation overwrites the outcome row
  double syy = 0.0;
  if (RL::LAST) {
#pragma unroll
    for (int j = 0; j < P; ++j) rhs[j] = LR(P, j);
    syy = LR(P, P);
  }
  if (ROLE == 0) LR(0, 0) = (double)n;
  bool ok = true;
  double dmin = __builtin_inf(), dmax = 0.0;
  double invd[NOWN > 0 ? NOWN : 1];
  // ---- factor: publishers 0 .. WR-2 in turn; L[i][j] = (A[i][j] - sum_{t<j} L[i][t] L[j][t]) invd[j], the sequence of fit_decide ----
  auto own_pivots = [&]() {  // everything to the left of the owned pivots is final
#pragma unroll
    for (int j = LO; j < HIP; ++j) {
      double d = LR(j, j);
#pragma unroll
      for (int t = 0; t < j; ++t) d = __builtin_fma(-LR(j, t), LR(j, t), d);
      ok = ok && (d > 0.0) && (d < __builtin_inf());
      const double ljj = __builtin_sqrt(d);
      LR(j, j) = ljj;
      invd[j - LO] = 1.0 / ljj;
      dmin = ljj < dmin ? ljj : dmin;
      dmax = ljj > dmax ? ljj : dmax;
#pragma unroll
      for (int i = j + 1; i < HI; ++i) {
        double vv = LR(i, j);
#pragma unroll
        for (int t = 0; t < j; ++t) vv = __builtin_fma(-LR(i, t), LR(j, t), vv);
        LR(i, j) = vv * invd[j - LO];
      }
    }
    if (!RL::LAST) {
#pragma unroll
      for (int k = 0; k < RL::NE; ++k) xb[k * WL] = acc[k];
#pragma unroll
      for (int j = 0; j < HI - LO; ++j) xb[(RL::NE + j) * WL] = invd[j];
      xb[(RL::NE + HI - LO + 0) * WL] = ok ? 1.0 : 0.0;
      xb[(RL::NE + HI - LO + 1) * WL] = dmin;
      xb[(RL::NE + HI - LO + 2) * WL] = dmax;
    }
  };
  auto consume = [&](auto tag) {  // complete the columns of publisher S in the owned rows
    constexpr int S = decltype(tag)::value;
    constexpr int SLO = wide_lo(V, S), SHI = wide_lo(V, S + 1), SNE = tri(SHI) - tri(SLO);
#pragma unroll
    for (int j = SLO; j < SHI; ++j) {
      double lj[P];
#pragma unroll
      for (int t = 0; t < j; ++t) lj[t] = xb[(tri(j) + t - tri(SLO)) * WL];
      const double ivd = xb[(SNE + j - SLO) * WL];
#pragma unroll
      for (int i = LO; i < HI; ++i) {
        double vv = LR(i, j);
#pragma unroll
        for (int t = 0; t < j; ++t) vv = __builtin_fma(-LR(i, t), lj[t], vv);
        LR(i, j) = vv * ivd;
      }
    }
    if (ROLE == S + 1) {  // the running rank statistics travel down the chain
      ok = xb[(SNE + SHI - SLO + 0) * WL] != 0.0;
      dmin = xb[(SNE + SHI - SLO + 1) * WL];
      dmax = xb[(SNE + SHI - SLO + 2) * WL];
    }
  };
  static_assert(WR == 4, "the publisher steps below are written out for four roles");
  if constexpr (ROLE == 0) own_pivots();
  __syncthreads();  // publisher 0 has written
  if constexpr (ROLE > 0) consume(std::integral_constant<int, 0>());
  __syncthreads();  // consumers are done with the chunk
  if constexpr (ROLE == 1) own_pivots();
  __syncthreads();
  if constexpr (ROLE > 1) consume(std::integral_constant<int, 1>());
  __syncthreads();
  if constexpr (ROLE == 2) own_pivots();
  __syncthreads();
  if constexpr (ROLE > 2) consume(std::integral_constant<int, 2>());
  __syncthreads();
  if constexpr (ROLE == 3) own_pivots();
  const bool conv = ok && (dmin / dmax >= EPS_RANK);  // meaningful in the last role

  // ---- solve L' beta = z (z = the outcome row of L, owned by the last role), from the last role up ----
  double beta[RL::LAST ? P : (NOWN > 0 ? NOWN : 1)];
#pragma unroll
  for (int s = WR - 1; s >= 0; --s) {
    if (s == ROLE) {
      double w[HIP > 0 ? HIP : 1];  // right-hand side of rows [0, HIP): z_i minus the terms of the roles below
#pragma unroll
      for (int i = 0; i < HIP; ++i) w[i] = RL::LAST ? LR(P, i) : xb[(P + i) * WL];
#pragma unroll
      for (int i = HIP - 1; i >= LO; --i) {
        double vv = w[i];
#pragma unroll
        for (int t = i + 1; t < HIP; ++t) vv = __builtin_fma(-LR(t, i), beta[RL::LAST ? t : t - LO], vv);
        beta[RL::LAST ? i : i - LO] = vv * invd[i - LO];
        xb[i * WL] = beta[RL::LAST ? i : i - LO];
      }
#pragma unroll
      for (int i = 0; i < LO; ++i) {
        double vv = w[i];
#pragma unroll
        for (int t = LO; t < HIP; ++t) vv = __builtin_fma(-LR(t, i), beta[RL::LAST ? t : t - LO], vv);
        xb[(P + i) * WL] = vv;
      }
    }
    __syncthreads();
  }
  double rss = 0.0, sst = 0.0, df = 1.0, sigma_sq = 0.0;
  if (RL::LAST) {
#pragma unroll
    for (int i = 0; i < LO; ++i) beta[i] = xb[i * WL];
    double bxty = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) bxty = __builtin_fma(beta[i], rhs[i], bxty);
    const double nd = (double)n;
    rss = syy - bxty;
    df = nd - (double)P;
    sigma_sq = rss / df;
    sst = syy - rhs[0] * rhs[0] / nd;
  }
  __syncthreads();  // beta has been read: the area turns over to the tests

  // ---- squared statistic per test: (c'beta)^2 / (sigma^2 ||L^-1 c||^2) (posthoc.rs:86-99), two tests per trip down the chain ----
  const uint32_t nt = prm.n_tests;
  const uint32_t n_groups = (nt + 1u) >> 1;
  for (uint32_t grp = 0; grp < n_groups; ++grp) {
#pragma unroll
    for (int s = 0; s < WR; ++s) {
      if (s == ROLE) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const uint32_t t = 2u * grp + (uint32_t)tt;
          if (t < nt) {
            const double* c = prm.contrast + t * MAX_P;
            double u[HIP > 0 ? HIP : 1];
            double nq = 0.0;
#pragma unroll
            for (int i = 0; i < LO; ++i) {
              u[i] = xb[(tt * P + i) * WL];
              if (RL::LAST) nq = __builtin_fma(u[i], u[i], nq);
            }
#pragma unroll
            for (int i = LO; i < HIP; ++i) {
              double a = c[i];
#pragma unroll
              for (int qq = 0; qq < i; ++qq) a = __builtin_fma(-LR(i, qq), u[qq], a);
              u[i] = a * invd[i - LO];
              if (RL::LAST) nq = __builtin_fma(u[i], u[i], nq);
              else xb[(tt * P + i) * WL] = u[i];
            }
            if (RL::LAST) {
              double est = 0.0;
#pragma unroll
              for (int i = 0; i < P; ++i) est = __builtin_fma(c[i], beta[i], est);
              const double se_sq = sigma_sq * nq;
              double ts = __builtin_nan("");
              if (conv && se_sq > FLOAT_NEAR_ZERO && se_sq < __builtin_inf()) ts = est * est / se_sq;
              xb[(2 * P + (int)t) * WL] = ts;
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (RL::LAST) {
    double tsq[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) tsq[t] = (uint32_t)t < nt ? xb[(2 * P + t) * WL] : __builtin_nan("");
    ols_finish<MT>(prm, tsq, conv, rss, sst, df, (double)(P - 1), g, active, out_row);
  }
#undef LR
}

template <int P, int K0, int K1>
struct WideMaxNE {
  static constexpr int value = (WideRole<P, K0>::NE > WideMaxNE<P, K0 + 1, K1>::value) ? WideRole<P, K0>::NE : WideMaxNE<P, K0 + 1, K1>::value;
};
template <int P, int K1>
struct WideMaxNE<P, K1, K1> {
  static constexpr int value = 0;
};

// A grid of sample sizes (GRID): the fit consumes the accumulators, so the launch is split like the narrow designs' (ols_sim_impl.hpp,
// SPLIT): the row kernel leaves every role's band of the triangle in HBM at each grid point -- prm.lmm_rec[(g * tri(V) + entry) *
// lmm_rec_stride + replicate], 153 doubles per fit at P = 16, coalesced across the replicates -- and ols_wide_fit_rec_kernel reloads
// them into the same four roles and runs wide_fit.
template <int P, int ROLE, int NACC>
__device__ __forceinline__ void wide_store(const SimParams& prm, const double (&acc)[NACC], uint32_t g, uint32_t local, bool active) {
  typedef WideRole<P, ROLE> RL;
  if (!active) return;
  double* p = prm.lmm_rec + ((size_t)g * tri(P + 1) + tri(RL::LO)) * prm.lmm_rec_stride + local;
#pragma unroll
  for (int k = 0; k < RL::NE; ++k) p[(size_t)k * prm.lmm_rec_stride] = acc[k];
}

template <int P, bool FACTORS, bool GRID>
// (the row-only kernel of a split launch fits 168 registers = three waves per SIMD at P = 14 and at P = 13 without factors; capped there,
// P = 15 / 16 spill 64-170 B and stay at two)
__global__ __launch_bounds__(WB, (GRID && (P == 14 || (P == 13 && !FACTORS))) ? 3 : 2) void ols_wide_kernel(const SimParams* __restrict__ pp) {
  constexpr int V = P + 1;                                   // variables: intercept, x_1 .. x_{P-1}, y
  constexpr int NACC = WideMaxNE<P, 0, WR>::value;
  constexpr int SLOTS = 4 * P;                               // 4 rows x (K + 1 <= P) draw columns
  extern __shared__ float wide_lds[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
  const uint32_t local = blockIdx.x * WL + lane;
  const bool active = local < prm.n_sims;
  const uint32_t lsim = active ? local : (prm.n_sims - 1u);
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);  // rng.rs:61-69
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t out_row = prm.out_base + lsim;
  const uint32_t n_grid = GRID ? prm.n_grid : 1u;
  const uint32_t n = prm.ns[n_grid - 1u], K = prm.K;  // rows of the largest sample size; the smaller ones are prefixes
  // LDS: rows area [row of the quad][variable 1 .. P][lane] f64 (= the exchange area of the fit: 64 doubles per lane),
  // the staging area [slot = 4 * column + row][lane] f32, then the waves' tail queues
  constexpr int ROWS_DOUBLES = 4 * P * WL > 64 * WL ? 4 * P * WL : 64 * WL;
  double* rows_d = reinterpret_cast<double*>(wide_lds);
  float* stage = wide_lds + 2 * ROWS_DOUBLES;
  uint32_t* tq = reinterpret_cast<uint32_t*>(stage + SLOTS * WL) + role * TAIL_QUEUE;
  double* xb = rows_d + lane;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  // one loop per role (wave-uniform dispatch).  Per block of 4 rows: (1) the role stages the Philox / inverse-CDF blocks
  // of ITS draw columns; barrier; (2) the role builds ONE row of the quad -- row `ROLE` -- with the shared bit-exact row
  // builder and leaves it in LDS widened to f64, so a row is built and converted once, not once per role; barrier;
  // (3) every role reads the four finished rows and adds them into ITS rows of the triangle.  Both areas are single-
  // buffered: (1) of the next block starts after every wave has passed barrier 2, (2) after every wave has passed
  // barrier 1 of that block, i.e. after its reads of the previous rows.
  auto rows = [&](auto tag) {
    constexpr int ROLE = decltype(tag)::value;
    RowCtx<P> cx;
    row_ctx_init<P, FACTORS>(prm, cx);
    typedef typename PendMask<P>::type mask_t;
    const uint32_t n_quads = (n + 3u) >> 2;
    float* st = stage + lane;
    uint32_t g = 0, next = GRID ? prm.ns[0] : 0xffffffffu;  // rows [0, next) make up grid point g
    for (uint32_t q = 0; q < n_quads; ++q) {
      mask_t pend = 0;
#pragma unroll
      for (int j = ROLE; j < P; j += WR) {  // draw column j: predictor j < K, the residual at j == K (rng.rs:141-147)
        if ((uint32_t)j <= K) {
          const u32x4 w = (uint32_t)j < K ? planar_words(k0, k1, CLASS_XNORM, (uint32_t)j, q) : planar_words(k0, k1, CLASS_RESID, 0u, q);
          stage_column<WL>(st, (uint32_t)j, w, pend);
        }
      }
      wide_fixup_tails<mask_t>(stage, lane, pend, tq);
      __syncthreads();  // (1) every column of the block is staged
      {
        const uint32_t rr = 4u * q + (uint32_t)ROLE;
        if (rr < n) {
          float x[P], y;
          build_row<P, FACTORS, WL, 0, false>(prm, cx, k0, k1, st, ROLE, rr, x, y, 0.0, nullptr);
          double* row = rows_d + (size_t)ROLE * P * WL + lane;
#pragma unroll
          for (int c = 1; c < P; ++c) row[(c - 1) * WL] = (double)x[c];
          row[(P - 1) * WL] = (double)y;
        }
      }
      __syncthreads();  // (2) the four rows of the block are built
#pragma unroll 2  // two rows in flight: the next row's LDS reads overlap this row's multiply-adds
      for (int r = 0; r < 4; ++r) {
        if (GRID && 4u * q + (uint32_t)r == next && g + 1u < n_grid) {  // every row of grid point g is in: park the band
          wide_store<P, ROLE>(prm, acc, g, local, active);
          ++g;
          next = prm.ns[g];
        }
        if (4u * q + (uint32_t)r < n) {
          const double* row = rows_d + (size_t)r * P * WL + lane;
          double v[V];
          v[0] = 1.0;
#pragma unroll
          for (int c = 1; c <= P; ++c) v[c] = row[(c - 1) * WL];
          WideRole<P, ROLE>::add_row(acc, v);
        }
      }
    }
    if (GRID) wide_store<P, ROLE>(prm, acc, n_grid - 1u, local, active);  // the largest sample size
  };
  if (role == 0u) rows(std::integral_constant<int, 0>());
  else if (role == 1u) rows(std::integral_constant<int, 1>());
  else if (role == 2u) rows(std::integral_constant<int, 2>());
  else rows(std::integral_constant<int, 3>());
  if (GRID) return;  // the fits run in ols_wide_fit_rec_kernel
  __syncthreads();  // the staging area becomes the exchange area of the fit
  // ---- the fit (batch.rs:956-1142), in place on the roles' rows; every role passes the same barriers ----
  if (role == 0u) wide_fit<P, 0>(prm, acc, xb, n, 0u, active, out_row);
  else if (role == 1u) wide_fit<P, 1>(prm, acc, xb, n, 0u, active, out_row);
  else if (role == 2u) wide_fit<P, 2>(prm, acc, xb, n, 0u, active, out_row);
  else wide_fit<P, 3>(prm, acc, xb, n, 0u, active, out_row);
}

// second half of a split wide launch: blockIdx.y = grid point; the four roles reload their bands and fit
template <int P>
__global__ __launch_bounds__(WB, 2) void ols_wide_fit_rec_kernel(const SimParams* __restrict__ pp) {
  constexpr int NACC = WideMaxNE<P, 0, WR>::value;
  extern __shared__ float wide_lds[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t local = blockIdx.x * WL + lane;
  const bool active = local < prm.n_sims;
  const uint32_t src = active ? local : (prm.n_sims - 1u);
  const uint32_t g = blockIdx.y;
  double* xb = reinterpret_cast<double*>(wide_lds) + lane;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  auto fit = [&](auto tag) {
    constexpr int ROLE = decltype(tag)::value;
    typedef WideRole<P, ROLE> RL;
    const double* p = prm.lmm_rec + ((size_t)g * tri(P + 1) + tri(RL::LO)) * prm.lmm_rec_stride + src;
#pragma unroll
    for (int k = 0; k < RL::NE; ++k) acc[k] = p[(size_t)k * prm.lmm_rec_stride];
    wide_fit<P, ROLE>(prm, acc, xb, prm.ns[g], g, active, prm.out_base + src);
  };
  if (role == 0u) fit(std::integral_constant<int, 0>());
  else if (role == 1u) fit(std::integral_constant<int, 1>());
  else if (role == 2u) fit(std::integral_constant<int, 2>());
  else fit(std::integral_constant<int, 3>());
}

}  // namespace

size_t ols_wide_lds_bytes(int P) {
  const size_t rows = (size_t)4 * P * WL * sizeof(double), exch = (size_t)64 * WL * sizeof(double);
  return (rows > exch ? rows : exch) + (size_t)4 * P * WL * sizeof(float) + WR * TAIL_QUEUE * sizeof(uint32_t);
}

size_t ols_wide_rec_fields(int P) { return (size_t)tri(P + 1); }

// P = 13 .. 16, plain data path (flavor 0 / 1), <= 16 tests: the host routes exactly these here; n_grid > 1 = the split launch
hipError_t launch_ols_wide(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, uint32_t n_grid, hipStream_t st) {
  const dim3 block(WB), grid((n_sims + WL - 1u) / WL);
  const size_t lds = ols_wide_lds_bytes(P);
  // P = 14 takes the split launch on one sample size too: its row-only kernel fits 168 registers (three waves per SIMD), the fused
  // one 195 (two) -- 6.9 -> 6.1 ms per 10^6 fits with TWO fits per replicate; at P = 15 / 16 both forms hold two waves and fused wins
  const bool split = n_grid > 1u || P == 14;
#define LWK(PP, FF, GG)                                                                                                 \
  {                                                                                                                     \
    if (lds > 48 * 1024) {                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)ols_wide_kernel<PP, FF, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                                    \
    }                                                                                                                   \
    hipLaunchKernelGGL((ols_wide_kernel<PP, FF, GG>), grid, block, lds, st, d_prm);                                     \
  }
#define LW(PP)                                                                                                          \
  case PP: {                                                                                                            \
    if (split) { if (flavor) LWK(PP, true, true) else LWK(PP, false, true) }                                            \
    else { if (flavor) LWK(PP, true, false) else LWK(PP, false, false) }                                                \
    if (split) {                                                                                                        \
      hipError_t e = hipGetLastError();                                                                                 \
      if (e != hipSuccess) return e;                                                                                    \
      const size_t flds = (size_t)64 * WL * sizeof(double);                                                             \
      hipLaunchKernelGGL((ols_wide_fit_rec_kernel<PP>), dim3((n_sims + WL - 1u) / WL, n_grid), block, flds, st, d_prm); \
    }                                                                                                                   \
  } break;
  switch (P) {
    LW(13) LW(14) LW(15) LW(16)
    default: return hipErrorInvalidValue;
  }
#undef LW
#undef LWK
  return hipGetLastError();
}

}  // namespace mcp
