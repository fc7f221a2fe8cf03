// mcp_scenario.hpp -- the general data-generating path on the device: per-replicate scenario design
// (perturbed correlation + PSD repair + Cholesky, marginal / residual swaps, per-study effects) and the
// f64 transform plane (Phi, marginals, t / chi-square residuals).
//
// Reference (crates/engine-core/src): scenarios.rs:50-186, data_gen.rs:990-1093 (populate_design),
// correlation.rs:82-190 (psd_repair_and_factor), distributions.rs:32-63 (phi), data_gen.rs:1106-1250
// (apply_marginal), marginals/t3.rs:82-96 (lookup), data_gen.rs:709-776 (beta jitter, hsk constants).
// PARITY: the reference composes glmm's owned exp / ln kernels (<= 2 ULP of libm) and faer's
// self_adjoint_eigen here; this path uses ocml exp / log and a cyclic Jacobi sweep, so it is pinned to
// the CPU oracle (same algorithm), not bit-pinned to the reference (SURVEY.md 8c "parity unpinned").
#pragma once
#include "mcp_device.hpp"
#include "mcp_params.hpp"

#pragma clang fp contract(off)

namespace mcp {

constexpr double EXP_CAP = 6.95925599364711;            // data_gen.rs:39-42
constexpr double EXP_CENSORED_MEAN = 0.9990501970288289;
constexpr double EXP_CENSORED_STD = 0.9933676327697134;
constexpr double SQRT3 = 1.7320508075688772;
constexpr double SCEN_NEAR_ZERO = 1e-30;

// sequential SimRng (rng.rs:51-128) for the cold scenario / heterogeneity streams
struct SeqRng {
  uint32_t k0, k1;
  uint64_t counter;
  u32x4 buf;
  uint32_t pos;
  __device__ __forceinline__ void init(uint64_t base_seed, uint64_t sim_id) {
    const uint64_t k = pcg_mix64(base_seed, sim_id);
    k0 = (uint32_t)k;
    k1 = (uint32_t)(k >> 32);
    counter = 0;
    pos = 4;
  }
  __device__ __forceinline__ uint32_t next_u32() {
    if (pos == 4) {
      buf = philox4x32_10((uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u, k0, k1);
      counter += 1;
      pos = 0;
    }
    const uint32_t w = pos == 0 ? buf.x : (pos == 1 ? buf.y : (pos == 2 ? buf.z : buf.w));
    pos += 1;
    return w;
  }
  __device__ __forceinline__ float next_uniform() { return u32_to_unit_f32(next_u32()); }
  __device__ __forceinline__ float next_normal() { return norm_inv_cdf_f32(next_uniform()); }
};

// Phi via Abramowitz-Stegun 7.1.26 erfc (distributions.rs:32-63), op for op
__device__ __forceinline__ double phi_as(double z) {
  const double x = -z * 0.70710678118654752440;
  const double sign = x < 0.0 ? -1.0 : 1.0;
  const double ax = __builtin_fabs(x);
  const double t = 1.0 / (1.0 + 0.3275911 * ax);
  const double y =
      1.0 - (((((1.061405429 * t + -1.453152027) * t) + 1.421413741) * t + -0.284496736) * t + 0.254829592) * t * exp(-ax * ax);
  return 0.5 * (1.0 - sign * y);
}

__device__ __forceinline__ double t3_lookup(const double* tab, double pct) {  // marginals/t3.rs:82-96
  const double pmin = 0.00121, pmax = 0.99879;
  const double p = pct < pmin ? pmin : (pct > pmax ? pmax : pct);
  const double scale = 2047.0 / (pmax - pmin);
  const double idx = (p - pmin) * scale;
  const uint32_t ii = (uint32_t)floor(idx);
  const double frac = idx - (double)ii;
  if (ii + 1u >= 2048u) return tab[2047];
  return tab[ii] * (1.0 - frac) + tab[ii + 1u] * frac;
}

// apply_marginal (data_gen.rs:1106-1182) on the f32-narrowed mix value widened to f64.
// Lanes of a wave carry different marginals (per-replicate swaps), so a switch with one Phi per arm makes the
// wave pay every arm.  Phi(z) and Phi(-z) share |x|, t and the exponential of the A&S form, so both are taken
// from ONE evaluation (bitwise what phi_as(z) / phi_as(-z) return), the two skewed shapes share ONE log, and
// the rest are selects: one exp + at most one log per value whatever mix of shapes the wave holds.
__device__ __forceinline__ double apply_marginal(uint32_t dist, double z, double param, const double* t3,
                                                 const double* upload_normal = nullptr, uint32_t upload_rows = 0u,
                                                 uint32_t upload_cols = 0u, uint32_t var_idx = 0u) {
  if (dist == 0u || dist > 7u) return z;
  const double x = -z * 0.70710678118654752440;
  const double ax = __builtin_fabs(x);
  const double t = 1.0 / (1.0 + 0.3275911 * ax);
  const double y =
      1.0 - (((((1.061405429 * t + -1.453152027) * t) + 1.421413741) * t + -0.284496736) * t + 0.254829592) * t * exp(-ax * ax);
  const double u_pos = 0.5 * (1.0 - (x < 0.0 ? -1.0 : 1.0) * y);  // phi_as(z)
  const double u_neg = 0.5 * (1.0 - (x > 0.0 ? -1.0 : 1.0) * y);  // phi_as(-z): its x is exactly -x
  double out = z;
  if (dist == 2u || dist == 3u) {  // Right / LeftSkewed: censored standardized Exp(1), mirrored
    double u = dist == 2u ? u_neg : u_pos;
    u = u > 1.0 / 2048.0 ? u : 1.0 / 2048.0;  // ln_owned's 2^-11 input floor
    double e = -log(u);
    e = e < EXP_CAP ? e : EXP_CAP;
    out = dist == 2u ? (e - EXP_CENSORED_MEAN) / EXP_CENSORED_STD : (EXP_CENSORED_MEAN - e) / EXP_CENSORED_STD;
  } else if (dist == 5u) {  // Uniform(-sqrt3, sqrt3)
    out = -SQRT3 + (SQRT3 - (-SQRT3)) * u_pos;
  } else if (dist == 1u || dist == 7u) {  // Binary / UploadedBinary: 1 in the high-z tail
    out = u_pos < 1.0 - param ? 0.0 : 1.0;
  } else if (dist == 4u) {  // HighKurtosis: standardized t(3) via the PPF table
    out = t3 ? t3_lookup(t3, u_pos) : z;
  } else if (dist == 6u) {  // UploadedData: NORTA, type-7 quantile of the sorted uploaded column (data_gen.rs:1253-1277)
    out = 0.0;
    if (upload_rows != 0u) {
      const double u = u_pos < 0.0 ? 0.0 : (u_pos > 1.0 ? 1.0 : u_pos);
      const double pos = u * (double)(upload_rows - 1u);
      const uint32_t lo = (uint32_t)floor(pos);
      const uint32_t hi = lo + 1u < upload_rows - 1u ? lo + 1u : upload_rows - 1u;
      const double frac = pos - (double)lo;
      const double val_lo = upload_normal[(size_t)lo * upload_cols + var_idx];
      const double val_hi = upload_normal[(size_t)hi * upload_cols + var_idx];
      out = val_lo + frac * (val_hi - val_lo);
    }
  }
  return out;
}

// cluster random effect c of a replicate (draw_residual on the sequential stream, data_gen.rs:146-161,
// :1289-1320): the c-th block of `re_draws_per_cluster` words; standardized to unit variance, scaled by tau outside
__device__ __forceinline__ double draw_cluster_effect(const SimParams& prm, uint32_t k0, uint32_t k1, uint32_t c) {
  const uint64_t base = (uint64_t)c * prm.re_draws_per_cluster;
  auto nrm = [&](uint64_t q) { return (double)norm_inv_cdf_f32(u32_to_unit_f32(sequential_word(k0, k1, q))); };
  const uint32_t dist = prm.lme_re_dist;
  if (dist == 0u) return nrm(base);
  if (dist == 4u) return (2.0 * (double)u32_to_unit_f32(sequential_word(k0, k1, base)) - 1.0) * SQRT3;
  const double df = prm.lme_re_df > 3.0 ? prm.lme_re_df : 3.0;
  long dfi = (long)floor(df + 0.5);  // llround for df >= 3
  dfi = dfi < 1 ? 1 : dfi;
  if (dist == 1u) {  // HighKurtosis: t(df) = z / sqrt(chi2_df / df), standardized
    const double z = nrm(base);
    double chi2 = 0.0;
    for (long i = 0; i < dfi; ++i) {
      const double g = nrm(base + 1u + (uint64_t)i);
      chi2 += g * g;
    }
    const double denom = __builtin_sqrt(chi2 / df);
    const double t = denom <= 0.0 ? z : z / denom;
    return t * (1.0 / __builtin_sqrt(df / (df - 2.0)));
  }
  double chi2 = 0.0;  // Right / LeftSkewed: centered scaled chi-square
  for (long i = 0; i < dfi; ++i) {
    const double g = nrm(base + (uint64_t)i);
    chi2 += g * g;
  }
  const double centered = (chi2 - df) * (1.0 / __builtin_sqrt(2.0 * df));
  return dist == 3u ? -centered : centered;
}

// lower Cholesky in place, column-major k x k (a[j*k+i] = row i, col j); false when a pivot is not positive
__device__ inline bool chol_colmajor(double* a, uint32_t k) {
  for (uint32_t j = 0; j < k; ++j) {
    double d = a[j * k + j];
    for (uint32_t t = 0; t < j; ++t) d -= a[t * k + j] * a[t * k + j];
    if (!(d > 0.0)) return false;
    const double ljj = __builtin_sqrt(d);
    a[j * k + j] = ljj;
    for (uint32_t i = j + 1; i < k; ++i) {
      double s = a[j * k + i];
      for (uint32_t t = 0; t < j; ++t) s -= a[t * k + i] * a[t * k + j];
      a[j * k + i] = s / ljj;
    }
  }
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t i = 0; i < j; ++i) a[j * k + i] = 0.0;
  return true;
}

// spectral PSD repair + factor (correlation.rs:82-190); cyclic Jacobi in place of faer's EVD
__device__ inline bool psd_repair_and_factor(double* m, uint32_t k) {
  double tmp[MAX_K * MAX_K];
  for (uint32_t i = 0; i < k * k; ++i) tmp[i] = m[i];
  if (chol_colmajor(tmp, k)) {
    for (uint32_t i = 0; i < k * k; ++i) m[i] = tmp[i];
    return true;
  }
  double a[MAX_K * MAX_K], v[MAX_K * MAX_K], w[MAX_K], rep[MAX_K * MAX_K], d[MAX_K];
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t i = 0; i < k; ++i) a[j * k + i] = (i >= j) ? m[j * k + i] : m[i * k + j];
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < k; ++j) v[j * k + i] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0;
    for (uint32_t i = 0; i < k; ++i)
      for (uint32_t j = i + 1; j < k; ++j) off += a[j * k + i] * a[j * k + i];
    if (off < 1e-30) break;
    for (uint32_t p = 0; p < k; ++p)
      for (uint32_t q = p + 1; q < k; ++q) {
        const double apq = a[q * k + p];
        if (__builtin_fabs(apq) < 1e-300) continue;
        const double theta = (a[q * k + q] - a[p * k + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (__builtin_fabs(theta) + __builtin_sqrt(theta * theta + 1.0));
        const double c = 1.0 / __builtin_sqrt(t * t + 1.0), s = t * c;
        for (uint32_t r = 0; r < k; ++r) {
          const double arp = a[p * k + r], arq = a[q * k + r];
          a[p * k + r] = c * arp - s * arq;
          a[q * k + r] = s * arp + c * arq;
        }
        for (uint32_t r = 0; r < k; ++r) {
          const double apr = a[r * k + p], aqr = a[r * k + q];
          a[r * k + p] = c * apr - s * aqr;
          a[r * k + q] = s * apr + c * aqr;
        }
        for (uint32_t r = 0; r < k; ++r) {
          const double vrp = v[p * k + r], vrq = v[q * k + r];
          v[p * k + r] = c * vrp - s * vrq;
          v[q * k + r] = s * vrp + c * vrq;
        }
      }
  }
  for (uint32_t i = 0; i < k; ++i) {
    w[i] = a[i * k + i];
    if (!(w[i] > 1e-10)) w[i] = 1e-10;  // PSD_EIGENVALUE_FLOOR (correlation.rs:20)
  }
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < k; ++j) {
      double acc = 0.0;
      for (uint32_t t = 0; t < k; ++t) acc += v[t * k + i] * w[t] * v[t * k + j];
      rep[j * k + i] = acc;
    }
  for (uint32_t i = 0; i < k; ++i) d[i] = rep[i * k + i] > 0.0 ? __builtin_sqrt(rep[i * k + i]) : 1.0;
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t i = 0; i < k; ++i) rep[j * k + i] /= d[i] * d[j];
  for (uint32_t i = 0; i < k; ++i) rep[i * k + i] = 1.0;
  for (uint32_t i = 0; i < k * k; ++i) m[i] = rep[i];
  return chol_colmajor(m, k);
}

}  // namespace mcp
