// ols_sim_impl.hpp -- the lane-per-replicate OLS simulate kernel and its launcher templates, included by the two
// translation units that instantiate it: ols_kernels.hip (continuous-only and factor designs, the staged pipeline) and
// ols_scen_kernels.hip (general / scenario data path, built with its own compiler flags -- see the Makefile).
#pragma once
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"
#include "mcp_suff.hpp"
#include "mcp_ols_finish.hpp"

#pragma clang fp contract(off)

namespace mcp {

enum : int { MODE_FUSED = 0, MODE_GENERATE = 1, MODE_FIT = 2 };

// Fit the accumulated prefix of n rows and decide every test (batch.rs:956-1142).
template <int P>
__device__ __forceinline__ void fit_decide(const SimParams& prm, const Suff<P>& s, uint32_t g, uint32_t n,
                                           bool active, uint32_t out_row) {
  // ---- X'X (lower) and X'y ----
  // P <= 8: the factor gets its own registers; P = 9..12 (MAX_P_FUSED): it overwrites X'X entry by entry -- an entry of
  // X'X is read for the last time when the factor's entry at the same place is formed -- so the live set stays one
  // triangle (78 doubles at P = 12) and the kernel keeps two waves per SIMD
  constexpr bool INPLACE = P > MAX_P_WAVE;
  constexpr int MT = P > MAX_P_WAVE ? MAX_P_FUSED : MAX_T_WAVE;  // tests per fit
  double A[P][P];
  double rhs[P];
  A[0][0] = (double)n;
  rhs[0] = s.sy;
  {
    int idx = 0;
#pragma unroll
    for (int a = 1; a < P; ++a) {
      A[a][0] = s.sx[a - 1];
      rhs[a] = s.sxy[a - 1];
#pragma unroll
      for (int b = 1; b <= a; ++b) A[a][b] = s.sxx[idx++];
    }
  }
  // ---- Cholesky, converged = min|L_ii| / max|L_ii| >= EPS_RANK (batch.rs:40-41) ----
  double Lsep[INPLACE ? 1 : P][INPLACE ? 1 : P];
#define Lm(i, j) (*(INPLACE ? &A[i][j] : &Lsep[INPLACE ? 0 : (i)][INPLACE ? 0 : (j)]))
  double invd[P];
  bool ok = true;
  double dmin = __builtin_inf(), dmax = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int t = 0; t < j; ++t) d = __builtin_fma(-Lm(j, t), Lm(j, t), d);
    ok = ok && (d > 0.0) && (d < __builtin_inf());
    const double ljj = __builtin_sqrt(d);
    Lm(j, j) = ljj;
    invd[j] = 1.0 / ljj;
    dmin = ljj < dmin ? ljj : dmin;
    dmax = ljj > dmax ? ljj : dmax;
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int t = 0; t < j; ++t) v = __builtin_fma(-Lm(i, t), Lm(j, t), v);
      Lm(i, j) = v * invd[j];
    }
  }
  const bool conv = ok && (dmin / dmax >= EPS_RANK);
  // ---- beta = (L L')^-1 X'y ----
  double zf[P], beta[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    double v = rhs[i];
#pragma unroll
    for (int t = 0; t < i; ++t) v = __builtin_fma(-Lm(i, t), zf[t], v);
    zf[i] = v * invd[i];
  }
#pragma unroll
  for (int i = P - 1; i >= 0; --i) {
    double v = zf[i];
#pragma unroll
    for (int t = i + 1; t < P; ++t) v = __builtin_fma(-Lm(t, i), beta[t], v);
    beta[i] = v * invd[i];
  }
  double bxty = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) bxty = __builtin_fma(beta[i], rhs[i], bxty);
  const double nd = (double)n;
  const double rss = s.syy - bxty;
  const double df = nd - (double)P;
  const double sigma_sq = rss / df;
  const double sst = s.syy - s.sy * s.sy / nd;

  // ---- squared statistic per test: (c'beta)^2 / (sigma^2 ||L^-1 c||^2)  (posthoc.rs:86-99) ----
  const uint32_t nt = prm.n_tests;
  double tsq[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    tsq[t] = __builtin_nan("");
    if ((uint32_t)t < nt) {
      const double* c = prm.contrast + t * MAX_P;
      double est = 0.0, nq = 0.0;
      double u[P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
        est = __builtin_fma(c[i], beta[i], est);
        double acc = c[i];
#pragma unroll
        for (int q = 0; q < i; ++q) acc = __builtin_fma(-Lm(i, q), u[q], acc);
        u[i] = acc * invd[i];
        nq = __builtin_fma(u[i], u[i], nq);
      }
      const double se_sq = sigma_sq * nq;
      if (conv && se_sq > FLOAT_NEAR_ZERO && se_sq < __builtin_inf()) tsq[t] = est * est / se_sq;
    }
  }

  // ---- decisions, per-replicate tensors, additive counters (mcp_ols_finish.hpp) ----
  ols_finish<MT>(prm, tsq, conv, rss, sst, df, (double)(P - 1), g, active, out_row);
#undef Lm
}

// sufficient statistics of grid point g, replicate slot `local`, between the two kernels of a split launch:
// prm.lmm_rec[(field * n_grid + g) * lmm_rec_stride + local], fields in Suff<P> order (coalesced across replicates)
template <int P>
__device__ __forceinline__ void suff_store(const SimParams& prm, const Suff<P>& s, uint32_t g, uint32_t local) {
  double* p = prm.lmm_rec + (size_t)g * prm.lmm_rec_stride + local;
  // the field stride stays opaque to the optimiser: as a loop invariant of the grid loop, the 27 products field * stride
  // were hoisted out of it and held in 50 VGPRs for the whole kernel
  uint32_t fs = prm.n_grid * prm.lmm_rec_stride;
  asm volatile("" : "+s"(fs));
#pragma unroll
  for (int i = 0; i < (P - 1) * P / 2; ++i) { *p = s.sxx[i]; p += fs; }
#pragma unroll
  for (int i = 0; i < P - 1; ++i) { *p = s.sx[i]; p += fs; }
#pragma unroll
  for (int i = 0; i < P - 1; ++i) { *p = s.sxy[i]; p += fs; }
  *p = s.sy; p += fs;
  *p = s.syy;
}
template <int P>
__device__ __forceinline__ void suff_load(const SimParams& prm, Suff<P>& s, uint32_t g, uint32_t local) {
  const double* base = prm.lmm_rec + (size_t)g * prm.lmm_rec_stride + local;
  const size_t fs = (size_t)prm.n_grid * prm.lmm_rec_stride;
  int f = 0;
#pragma unroll
  for (int i = 0; i < (P - 1) * P / 2; ++i) s.sxx[i] = base[(size_t)(f++) * fs];
#pragma unroll
  for (int i = 0; i < P - 1; ++i) s.sx[i] = base[(size_t)(f++) * fs];
#pragma unroll
  for (int i = 0; i < P - 1; ++i) s.sxy[i] = base[(size_t)(f++) * fs];
  s.sy = base[(size_t)(f++) * fs];
  s.syy = base[(size_t)(f++) * fs];
}

// tile layout of the staged pipeline: [block of 256 replicates][column: x1..x_{P-1}, y][row quad][256 lanes][4 rows]
// f32 -- a lane owns 16 contiguous bytes per (column, row quad), so generator stores and fit loads are
// dwordx4 (1 KiB per wave instruction) and a workgroup's whole design is one contiguous region.
__device__ __forceinline__ size_t tile_quad_index(uint32_t P, uint32_t n_quads, uint32_t blk, uint32_t c, uint32_t quad) {
  return ((((size_t)blk * P + c) * n_quads + quad) * BLOCK) * 4u;  // + tid * 4 + r
}

// MODE_FUSED   : generate + accumulate + fit, nothing but flags/counters leaves the chip
// MODE_GENERATE: generate and store the design tile to HBM
// MODE_FIT     : stream the tile back (coalesced across replicates) and fit
#ifndef MCP_LB_W
#define MCP_LB_W 1
#endif
#ifndef MCP_LB_SCEN
#define MCP_LB_SCEN 2  // row kernel of a split general-path launch, P <= 7: waves per SIMD the register allocator leaves room for
#endif

// GRID1: a single sample size (the common find_power call): the grid loop disappears and the sufficient
// statistics are dead once X'X is formed, which frees their registers for the solve.
// FLAVOR 0: continuous-only design (the factor walk / categorical code is compiled out);
// FLAVOR 1: + factor dummies; FLAVOR 2: general path (per-replicate scenario design, non-normal marginals /
// residuals, per-study effects, heteroskedastic scaling) on top of FLAVOR 1.
template <int P, int MODE, bool GRID1, int FLAVOR>
__global__ __launch_bounds__(BLOCK, (FLAVOR == 2 && MODE == MODE_FUSED && P <= 7) ? MCP_LB_SCEN : MCP_LB_W) void ols_sim_kernel(const SimParams* __restrict__ pp) {
  constexpr bool FACTORS = FLAVOR >= 1;
  constexpr bool SCEN = FLAVOR == 2;
  // General path and every call with a grid of sample sizes, narrow designs: the launch is SPLIT -- this kernel leaves the sufficient statistics of every grid point in
  // HBM (prm.lmm_rec, 27 doubles per fit at P = 6) and ols_fit_rec_kernel runs the fits.  Inline, the fit's 160 registers
  // on top of the accumulators and the per-replicate scenario record put the kernel at 280 VGPRs = one wave per SIMD, and
  // its 20 KB of code between the row loops of two grid points; split, the row kernel keeps three waves per SIMD.
  constexpr bool SPLIT = MODE == MODE_FUSED && P <= MAX_P_WAVE && (SCEN || !GRID1);
  extern __shared__ float stage_lds[];
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = local < prm.n_sims;
  // general path, fused mode: lanes take the replicates of the launch in transform-class order (scenario_kernels.hip)
  const uint32_t lsim = active ? ((SCEN && MODE == MODE_FUSED && prm.scen_perm) ? prm.scen_perm[local] : local) : (prm.n_sims - 1u);
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);  // rng.rs:61-69
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t out_row = prm.out_base + lsim;
  const uint32_t max_n = prm.max_n;
  float* st = stage_lds + threadIdx.x;
  // dense tail queue of this wave, behind the staging area
  uint32_t* tq = reinterpret_cast<uint32_t*>(stage_lds + (size_t)4 * (P + (SCEN ? 1 : 0)) * BLOCK) + (threadIdx.x >> 6) * TAIL_QUEUE;
  RowCtx<P> cx;
  if (MODE != MODE_FIT) row_ctx_init<P, FACTORS>(prm, cx);
  ScenLane<P> sl;
  if (SCEN && MODE != MODE_FIT) scen_lane_load<P>(prm, lsim, sl);

  const uint32_t n_quads = (max_n + 3u) >> 2;
  if (MODE == MODE_GENERATE) {
    for (uint32_t ctr = 0; ctr < n_quads; ++ctr) {
      stage_group<P, BLOCK, 0, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u, tq);
      float q[P][4];  // [column: x1..x_{P-1}, y][row of the quad]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x[P], y;
        build_row<P, FACTORS, BLOCK, 0, SCEN>(prm, cx, k0, k1, st, r, 4u * ctr + r, x, y, 0.0, &sl);
#pragma unroll
        for (int c = 1; c < P; ++c) q[c - 1][r] = x[c];
        q[P - 1][r] = y;
      }
#pragma unroll
      for (int c = 0; c < P; ++c)
        *reinterpret_cast<float4*>(prm.tile + tile_quad_index(P, n_quads, blockIdx.x, c, ctr) + threadIdx.x * 4u) =
            make_float4(q[c][0], q[c][1], q[c][2], q[c][3]);
    }
    return;
  }

  Suff<P> s;
  suff_reset(s);
  uint32_t row = 0;
  const uint32_t n_grid = GRID1 ? 1u : prm.n_grid;
  for (uint32_t g = 0; g < n_grid; ++g) {
    const uint32_t n_end = prm.ns[g];
    if (MODE == MODE_FUSED) {
      while (row < n_end) {
        const uint32_t base = row & ~3u;
        if ((row & 3u) == 0u) stage_group<P, BLOCK, 0, SCEN>(prm, k0, k1, row >> 2, st, SCEN ? sl.meta : 0u, tq);
        if constexpr (SCEN) {
          // general path: the correlated mix and the marginal transforms of the whole quad run in place in the staging area,
          // the transform as ONE run-time loop over its 4 K values (scen_premix_quad) -- inlined per column and row it is 20
          // copies of a 1.2 KB sequence and the row loop no longer fits the 64 KB instruction cache two CUs share
          // (profiles/r03_c_c5real_*: one VALU instruction per 11 cycles per wave whatever the occupancy)
          if ((row & 3u) == 0u) scen_premix_quad<P, BLOCK>(prm, st, sl);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t rr = base + (uint32_t)r;
            if (rr >= row && rr < n_end) {
              float x[P], y;
              const float e = build_x<P, FACTORS, BLOCK, SCEN, true>(prm, cx, k0, k1, st, r, rr, x, &sl);
              finish_row<P, 0, SCEN>(prm, cx, x, e, y, 0.0, &sl);
              suff_add_row<P>(s, x, y);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t rr = base + (uint32_t)r;
            if (rr >= row && rr < n_end) {
              float x[P], y;
              build_row<P, FACTORS, BLOCK, 0, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, 0.0, &sl);
              suff_add_row<P>(s, x, y);
            }
          }
        }
        const uint32_t nxt = base + 4u;
        row = nxt < n_end ? nxt : n_end;
      }
    } else {  // MODE_FIT: dwordx4 loads, lane = replicate; whole row quads take the vector path
      while (row < n_end) {
        if ((row & 3u) == 0u && row + 4u <= n_end) {
          typedef float f32x4 __attribute__((ext_vector_type(4)));
          f32x4 v[P];
#pragma unroll
          for (int c = 0; c < P; ++c)
            v[c] = __builtin_nontemporal_load(
                reinterpret_cast<const f32x4*>(prm.tile + tile_quad_index(P, n_quads, blockIdx.x, c, row >> 2) + threadIdx.x * 4u));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x[P];
#pragma unroll
            for (int c = 1; c < P; ++c) x[c] = r == 0 ? v[c - 1].x : (r == 1 ? v[c - 1].y : (r == 2 ? v[c - 1].z : v[c - 1].w));
            const float y = r == 0 ? v[P - 1].x : (r == 1 ? v[P - 1].y : (r == 2 ? v[P - 1].z : v[P - 1].w));
            suff_add_row<P>(s, x, y);
          }
          row += 4u;
        } else {  // ragged grid boundary: one row
          float x[P];
#pragma unroll
          for (int c = 1; c < P; ++c)
            x[c] = prm.tile[tile_quad_index(P, n_quads, blockIdx.x, c - 1, row >> 2) + threadIdx.x * 4u + (row & 3u)];
          const float y = prm.tile[tile_quad_index(P, n_quads, blockIdx.x, P - 1, row >> 2) + threadIdx.x * 4u + (row & 3u)];
          suff_add_row<P>(s, x, y);
          row += 1u;
        }
      }
    }
    if constexpr (SPLIT) {
      if (active) suff_store<P>(prm, s, g, local);
    } else {
      fit_decide<P>(prm, s, g, n_end, active, out_row);
    }
  }
}

// General path, second half of a split launch: one lane per (replicate, grid point) reloads the sufficient statistics the
// row kernel left in prm.lmm_rec and runs the fit the fused kernels run inline -- the same fit_decide, so the same bits.
template <int P>
__global__ __launch_bounds__(BLOCK) void ols_fit_rec_kernel(const SimParams* __restrict__ pp) {
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = local < prm.n_sims;
  const uint32_t src = active ? local : (prm.n_sims - 1u);
  const uint32_t lsim = prm.scen_perm ? prm.scen_perm[src] : src;
  const uint32_t g = blockIdx.y;
  Suff<P> s;
  suff_load<P>(prm, s, g, src);
  fit_decide<P>(prm, s, g, prm.ns[g], active, prm.out_base + lsim);
}

}  // namespace mcp

// ---- launcher templates: each translation unit instantiates the flavours it owns ----
namespace mcp {

hipError_t launch_ols_wide(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, uint32_t n_grid, hipStream_t st);  // ols_wide_kernels.hip

// fused kernel, P = 9..12 (MAX_P_FUSED): the staging area passes 48 KB at P = 12
template <bool GRID1, int FLAVOR>
static hipError_t launch_fused_wide(int P, const SimParams* d_prm, uint32_t n_sims, size_t lds, hipStream_t st, uint32_t n_grid) {
  const dim3 block(BLOCK), grid((n_sims + BLOCK - 1u) / BLOCK);
#define LFW(PP)                                                                                                         \
  case PP: {                                                                                                            \
    if (lds > 48 * 1024) {                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)ols_sim_kernel<PP, MODE_FUSED, GRID1, FLAVOR>,                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      if (e != hipSuccess) return e;                                                                                    \
    }                                                                                                                   \
    hipLaunchKernelGGL((ols_sim_kernel<PP, MODE_FUSED, GRID1, FLAVOR>), grid, block, lds, st, d_prm);                   \
  } break;
  switch (P) {
    LFW(9) LFW(10) LFW(11) LFW(12)
    // P = 13 .. 16, plain data path: four waves share 64 replicates (ols_wide_kernels.hip); a grid of sizes = its split launch
    default: return (FLAVOR != 2 && P >= 13 && P <= MAX_P) ? launch_ols_wide(P, d_prm, n_sims, FLAVOR, GRID1 ? 1u : n_grid, st) : hipErrorInvalidValue;
  }
#undef LFW
  return hipGetLastError();
}

template <int MODE, bool GRID1, int FLAVOR>
static hipError_t launch_sim_pgf(int P, const SimParams* d_prm, uint32_t n_sims, hipStream_t st, uint32_t n_grid = 1u) {
  const dim3 block(BLOCK), grid((n_sims + BLOCK - 1u) / BLOCK);
  // staging area: 4 rows x (K + 1 <= P) columns x 256 lanes of f32 (+1 scratch column on the general path;
  // none for the fit-only kernel)
  const size_t lds = MODE == MODE_FIT ? 0 : (size_t)4 * (P + (FLAVOR == 2 ? 1 : 0)) * BLOCK * sizeof(float) +
                                                (BLOCK / 64) * TAIL_QUEUE * sizeof(uint32_t);
  switch (P) {
    case 2: hipLaunchKernelGGL((ols_sim_kernel<2, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 3: hipLaunchKernelGGL((ols_sim_kernel<3, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 4: hipLaunchKernelGGL((ols_sim_kernel<4, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 5: hipLaunchKernelGGL((ols_sim_kernel<5, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 6: hipLaunchKernelGGL((ols_sim_kernel<6, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 7: hipLaunchKernelGGL((ols_sim_kernel<7, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 8: hipLaunchKernelGGL((ols_sim_kernel<8, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    default:
      if (MODE == MODE_FUSED) return launch_fused_wide<GRID1, FLAVOR>(P, d_prm, n_sims, lds, st, n_grid);
      return hipErrorInvalidValue;
  }
  if constexpr (MODE == MODE_FUSED && (FLAVOR == 2 || !GRID1)) {  // split launch (ols_sim_kernel, SPLIT): the fits of every (replicate, grid point)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 fgrid((n_sims + BLOCK - 1u) / BLOCK, n_grid);
    switch (P) {
      case 2: hipLaunchKernelGGL((ols_fit_rec_kernel<2>), fgrid, block, 0, st, d_prm); break;
      case 3: hipLaunchKernelGGL((ols_fit_rec_kernel<3>), fgrid, block, 0, st, d_prm); break;
      case 4: hipLaunchKernelGGL((ols_fit_rec_kernel<4>), fgrid, block, 0, st, d_prm); break;
      case 5: hipLaunchKernelGGL((ols_fit_rec_kernel<5>), fgrid, block, 0, st, d_prm); break;
      case 6: hipLaunchKernelGGL((ols_fit_rec_kernel<6>), fgrid, block, 0, st, d_prm); break;
      case 7: hipLaunchKernelGGL((ols_fit_rec_kernel<7>), fgrid, block, 0, st, d_prm); break;
      case 8: hipLaunchKernelGGL((ols_fit_rec_kernel<8>), fgrid, block, 0, st, d_prm); break;
      default: break;
    }
  }
  return hipGetLastError();
}

}  // namespace mcp
