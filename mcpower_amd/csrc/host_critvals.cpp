// host_critvals.cpp -- once-per-batch critical values in squared space (host side).
//
// Restates /root/reference/crates/engine-core/src/critvals.rs: norm_ppf :33 (Acklam),
// betai/betacf :91-186 (NR 6.4 Lentz), ln_gamma :112 (Lanczos g=7,n=9), t_ppf :211
// (Cornish-Fisher seed + Newton on the incomplete-beta CDF), gamma_inc_p :294,
// chi2_ppf :356 (Wilson-Hilferty + Newton), f_ppf :450, and
// CritValueTable::build_with_tukey_k :827-1012.  The per-replicate path never
// evaluates a quantile: it only compares t^2 with these thresholds (critvals.rs:3-6).
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/mcpower_hip.h"
#include "host_internal.hpp"

namespace mcp {
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();
constexpr double kPi = 3.14159265358979323846;

double ln_gamma(double x) {
  static const double c[9] = {0.9999999999998099,   676.5203681218851,    -1259.1392167224028,
                              771.3234287776531,    -176.6150291621406,   12.507343278686905,
                              -0.13857109526572012, 9.984369578019572e-6, 1.5056327351493116e-7};
  if (x < 0.5) return std::log(kPi / std::sin(kPi * x)) - ln_gamma(1.0 - x);
  x -= 1.0;
  double a = c[0];
  const double t = x + 7.5;
  for (int i = 1; i < 9; ++i) a += c[i] / (x + i);
  return 0.5 * std::log(2.0 * kPi) + (x + 0.5) * std::log(t) - t + std::log(a);
}

double ln_beta(double a, double b) { return ln_gamma(a) + ln_gamma(b) - ln_gamma(a + b); }

double clampmin(double v) { return std::fabs(v) < 1.0e-300 ? 1.0e-300 : v; }

double betacf(double a, double b, double x) {
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0;
  double d = 1.0 / clampmin(1.0 - qab * x / qap);
  double h = d;
  for (int m = 1; m <= 200; ++m) {
    const double mf = m, m2 = 2.0 * mf;
    double aa = mf * (b - mf) * x / ((qam + m2) * (a + m2));
    d = 1.0 / clampmin(1.0 + aa * d);
    c = clampmin(1.0 + aa / c);
    h *= d * c;
    aa = -(a + mf) * (qab + mf) * x / ((a + m2) * (qap + m2));
    d = 1.0 / clampmin(1.0 + aa * d);
    c = clampmin(1.0 + aa / c);
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 3e-15) break;
  }
  return h;
}

double betai(double a, double b, double x) {
  if (x <= 0.0) return 0.0;
  if (x >= 1.0) return 1.0;
  const double bt = std::exp(a * std::log(x) + b * std::log(1.0 - x) - ln_beta(a, b));
  return x < (a + 1.0) / (a + b + 2.0) ? bt * betacf(a, b, x) / a : 1.0 - bt * betacf(b, a, 1.0 - x) / b;
}

double t_cdf(double x, double df) {
  if (!std::isfinite(x)) return x > 0.0 ? 1.0 : 0.0;
  const double half = 0.5 * betai(0.5 * df, 0.5, df / (df + x * x));
  return x >= 0.0 ? 1.0 - half : half;
}

double gamma_inc_p(double a, double x) {
  if (x <= 0.0 || a <= 0.0) return 0.0;
  const double pre = std::exp(-x + a * std::log(x) - ln_gamma(a));
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, term = sum;
    for (int i = 0; i < 200; ++i) {
      ap += 1.0;
      term *= x / ap;
      sum += term;
      if (std::fabs(term) < std::fabs(sum) * 3e-15) break;
    }
    return sum * pre;
  }
  double b = x + 1.0 - a, c = 1.0 / 1.0e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 200; ++i) {
    const double an = -double(i) * (double(i) - a);
    b += 2.0;
    d = 1.0 / clampmin(an * d + b);
    c = clampmin(b + an / c);
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 3e-15) break;
  }
  return 1.0 - pre * h;
}

double wilson_hilferty(double z, double k) {
  const double h = 2.0 / (9.0 * k);
  const double b = 1.0 - h + z * std::sqrt(h);
  return k * (b * b * b);
}

// Newton on a monotone CDF with positivity guard (chi2_ppf :385-405, f_ppf :467-484)
template <class Cdf, class Pdf>
double newton_positive(double x, double p, Cdf cdf, Pdf pdf) {
  for (int it = 0; it < 80; ++it) {
    const double dens = pdf(x);
    if (!std::isfinite(dens) || dens <= 0.0) break;
    double nx = x - (cdf(x) - p) / dens;
    if (!std::isfinite(nx) || nx <= 0.0) nx = 0.5 * x;
    const double rel = std::fabs(nx - x) / (1.0 + std::fabs(x));
    x = nx;
    if (rel < 1e-13) break;
  }
  return x;
}

}  // namespace

double norm_ppf(double p) {
  static const double A[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                              1.38357751867269e2,     -3.066479806614716e+01, 2.506628277459239e+00};
  static const double B[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                              6.680131188771972e+01, -1.328068155288572e+01};
  static const double Cc[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
  static const double D[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                              3.754408661907416e+00};
  if (p <= 0.0) return -kInf;
  if (p >= 1.0) return kInf;
  auto tail = [&](double q) {
    return (((((Cc[0] * q + Cc[1]) * q + Cc[2]) * q + Cc[3]) * q + Cc[4]) * q + Cc[5]) /
           ((((D[0] * q + D[1]) * q + D[2]) * q + D[3]) * q + 1.0);
  };
  if (p < 0.02425) return tail(std::sqrt(-2.0 * std::log(p)));
  if (p <= 1.0 - 0.02425) {
    const double q = p - 0.5, r = q * q;
    return (((((A[0] * r + A[1]) * r + A[2]) * r + A[3]) * r + A[4]) * r + A[5]) * q /
           (((((B[0] * r + B[1]) * r + B[2]) * r + B[3]) * r + B[4]) * r + 1.0);
  }
  return -tail(std::sqrt(-2.0 * std::log(1.0 - p)));
}

double t_ppf(double p, double df) {
  if (!std::isfinite(df) || df <= 0.0) return kNaN;
  if (p <= 0.0) return -kInf;
  if (p >= 1.0) return kInf;
  if (p == 0.5) return 0.0;
  const bool upper = p > 0.5;
  const double q = upper ? p : 1.0 - p;
  double x;
  if (df > 200.0) {
    const double z = norm_ppf(q), z2 = z * z;
    x = z + (z * (z2 + 1.0)) / (4.0 * df) + (z * (5.0 * z2 * z2 + 16.0 * z2 + 3.0)) / (96.0 * df * df);
  } else if (std::fabs(df - 1.0) < 1e-9) {
    x = std::fabs(std::tan(kPi * (q - 0.5)));
  } else if (std::fabs(df - 2.0) < 1e-9) {
    x = std::sqrt(2.0 / (4.0 * q * (1.0 - q)) - 2.0) * (q > 0.5 ? 1.0 : -1.0);
  } else {
    x = norm_ppf(q);
  }
  if (df <= 1.0e5) {  // beyond that the incomplete-beta CDF loses accuracy; keep the seed (critvals.rs:247-254)
    const double lognorm = ln_gamma(0.5 * (df + 1.0)) - ln_gamma(0.5 * df) - 0.5 * std::log(df * kPi);
    for (int it = 0; it < 80; ++it) {
      const double pdf = std::exp(lognorm - 0.5 * (df + 1.0) * std::log(1.0 + x * x / df));
      if (pdf <= 0.0 || !std::isfinite(pdf)) break;
      const double nx = x - (t_cdf(x, df) - q) / pdf;
      if (!std::isfinite(nx)) break;
      const double rel = std::fabs(nx - x) / (1.0 + std::fabs(x));
      x = nx;
      if (rel < 1e-14) break;
    }
  }
  return upper ? x : -x;
}

double chi2_ppf(double p, double k) {
  if (!std::isfinite(k) || k <= 0.0) return kNaN;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return kInf;
  double x = wilson_hilferty(norm_ppf(p), k);
  if (!std::isfinite(x) || x <= 0.0) x = k > 1e-6 ? k : 1e-6;
  const double a = 0.5 * k, lg = ln_gamma(a);
  return newton_positive(
      x, p, [&](double v) { return gamma_inc_p(a, 0.5 * v); },
      [&](double v) { return std::exp((a - 1.0) * std::log(0.5 * v) - 0.5 * v - 0.69314718055994530942 - lg); });
}

double f_ppf(double p, double dfn, double dfd) {
  if (!std::isfinite(dfn) || !std::isfinite(dfd) || dfn <= 0.0 || dfd <= 0.0) return kNaN;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return kInf;
  double x = wilson_hilferty(norm_ppf(p), dfn) / dfn;
  if (!(x > 1e-6)) x = 1e-6;
  const double a = 0.5 * dfn, b = 0.5 * dfd, lb = ln_beta(a, b);
  return newton_positive(
      x, p,
      [&](double v) {
        if (v <= 0.0) return 0.0;
        if (!std::isfinite(v)) return 1.0;
        return 1.0 - betai(b, a, dfd / (dfd + dfn * v));
      },
      [&](double v) {
        if (v <= 0.0) return 0.0;
        return std::exp(a * std::log(dfn / dfd) + (a - 1.0) * std::log(v) - (a + b) * std::log(1.0 + dfn / dfd * v) - lb);
      });
}

// standardized t(3) PPF table of the HighKurtosis marginal (marginals/t3.rs:44-80): 2048 knots on
// [0.00121, 0.99879], scaled to unit variance by the trapezoid moments of the clamped table
std::vector<double> build_t3_table() {
  const int n = 2048;
  const double pmin = 0.00121, pmax = 0.99879, denom = n - 1;
  std::vector<double> v(n);
  for (int i = 0; i < n; ++i) v[i] = t_ppf(pmin + (pmax - pmin) * double(i) / denom, 3.0);
  const double du = (pmax - pmin) / denom;
  double m1 = pmin * v[0] + (1.0 - pmax) * v[n - 1];
  double m2 = pmin * v[0] * v[0] + (1.0 - pmax) * v[n - 1] * v[n - 1];
  for (int i = 0; i + 1 < n; ++i) {
    const double a = v[i], b = v[i + 1];
    m1 += du * (a + b) / 2.0;
    m2 += du * (a * a + a * b + b * b) / 3.0;
  }
  const double sd = std::sqrt(m2 - m1 * m1);
  for (double& x : v) x /= sd;
  return v;
}

// ---- studentized range (Tukey HSD thresholds), critvals.rs:486-760 ---------------------------------
// The normal CDF inside the range integral is the engine's own Abramowitz-Stegun erfc form
// (distributions.rs:32-56), so thresholds carry the same ~1e-7 bias as the reference's.
namespace {
const double GL32_X[16] = {0.04830766568773831, 0.1444719615827965, 0.23928736225213706, 0.33186860228212767,
                           0.42135127613063533, 0.5068999089322294, 0.5877157572407623,  0.6630442669302152,
                           0.7321821187402897,  0.7944837959679424, 0.84936761373257,    0.8963211557660522,
                           0.9349060759377397,  0.9647622555875064, 0.9856115115452684,  0.9972638618494816};
const double GL32_W[16] = {0.09654008851472781,  0.09563872007927483,  0.09384439908080457,  0.09117387869576386,
                           0.08765209300440391,  0.08331192422694685,  0.07819389578707031,  0.07234579410884845,
                           0.06582222277636175,  0.058684093478535704, 0.050998059262376244, 0.042835898022226426,
                           0.034273862913021626, 0.025392065309262427, 0.016274394730905965, 0.007018610009469298};

double phi_as(double z) {
  const double x = -z * 0.70710678118654752440;
  const double sign = x < 0.0 ? -1.0 : 1.0;
  const double ax = std::fabs(x);
  const double t = 1.0 / (1.0 + 0.3275911 * ax);
  const double y =
      1.0 - (((((1.061405429 * t + -1.453152027) * t) + 1.421413741) * t + -0.284496736) * t + 0.254829592) * t * std::exp(-ax * ax);
  return 0.5 * (1.0 - sign * y);
}

double norm_pdf(double z) { return std::exp(-0.5 * z * z) / std::sqrt(2.0 * kPi); }

// P(range of k iid N(0,1) <= w): k * int_{-8}^{8} phi(z) [Phi(z) - Phi(z - w)]^(k-1) dz, 4 + (k-1) GL32 panels
double range_cdf(double w, double k) {
  if (w <= 0.0) return 0.0;
  const uint64_t ki = (uint64_t)k;
  const uint64_t panels = 4 + (ki > 0 ? ki - 1 : 0);
  const double step = 16.0 / (double)panels, half = 0.5 * step;
  double acc = 0.0;
  auto f = [&](double z) {
    const double inner = phi_as(z) - phi_as(z - w);
    return inner > 0.0 ? norm_pdf(z) * std::pow(inner, k - 1.0) : 0.0;
  };
  for (uint64_t pn = 0; pn < panels; ++pn) {
    const double mid = -8.0 + ((double)pn + 0.5) * step;
    for (int i = 0; i < 16; ++i) {
      const double dx = half * GL32_X[i];
      acc += GL32_W[i] * (f(mid + dx) + f(mid - dx));
    }
  }
  const double v = k * (half * acc);
  return v < 1.0 ? v : 1.0;
}

// P(W <= q | k, df): gamma mixing in u = chi2_df / df on a mesh graded toward 0 (critvals.rs:640-673)
double studentized_range_cdf(double q, double k, double df) {
  if (q <= 0.0) return 0.0;
  if (!std::isfinite(df) || df > 2.0e4) return range_cdf(q, k);
  const double half_df = 0.5 * df;
  const double ln_c = half_df * std::log(half_df) - ln_gamma(half_df);
  const double u_max = 1.0 + 12.0 * std::sqrt(2.0 / df);
  const int panels = df < 6.0 ? 96 : 24;
  const double grade = df < 6.0 ? 3.0 : 2.0;
  auto g = [&](double u) {
    if (u <= 0.0) return 0.0;
    return std::exp(ln_c + (half_df - 1.0) * std::log(u) - half_df * u) * range_cdf(q * std::sqrt(u), k);
  };
  double acc = 0.0, a = 0.0;
  for (int pn = 1; pn <= panels; ++pn) {
    const double b = u_max * std::pow((double)pn / (double)panels, grade);
    const double half = 0.5 * (b - a), mid = 0.5 * (b + a);
    for (int i = 0; i < 16; ++i) {
      const double dx = half * GL32_X[i];
      acc += half * GL32_W[i] * (g(mid + dx) + g(mid - dx));
    }
    a = b;
  }
  return acc < 1.0 ? acc : 1.0;
}
}  // namespace

// critvals.rs:687-760: Bonferroni seed, Newton with a central-difference slope, bisection fallback
double q_tukey_ppf(double p, double k, double df) {
  if (!std::isfinite(k) || k <= 1.0 || !std::isfinite(df) || df < 2.0 || !std::isfinite(p)) return kNaN;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return kInf;
  const double m = k * (k - 1.0);
  const double t_q = t_ppf(1.0 - (1.0 - p) / m, df);
  double q = (std::isfinite(t_q) && t_q > 0.0) ? std::sqrt(2.0) * t_q : 3.0;
  double lo = 0.0, hi = (q > 1.0 ? q : 1.0) * 2.0 + 4.0;
  while (studentized_range_cdf(hi, k, df) < p && hi < 1.0e3) hi *= 1.5;
  const double h = 1e-4;
  for (int it = 0; it < 60; ++it) {
    const double f = studentized_range_cdf(q, k, df) - p;
    if (std::fabs(f) < 1e-9) return q;
    if (f < 0.0) lo = q; else hi = q;
    const double deriv = (studentized_range_cdf(q + h, k, df) - studentized_range_cdf(q - h, k, df)) / (2.0 * h);
    double next = (std::isfinite(deriv) && deriv > 1e-12) ? q - f / deriv : kNaN;
    if (!std::isfinite(next) || next <= lo || next >= hi) next = 0.5 * (lo + hi);
    if (std::fabs(next - q) < 1e-10) return next;
    q = next;
  }
  return q;
}

// memo of (alpha, k, df) -> (q / sqrt 2)^2: a grid revisits the same df for many rows, and one quantile costs
// ~10^5 range-integral evaluations
double TukeyMemo::threshold(double alpha, double k, double df) {
  for (const auto& e : entries)
    if (e.alpha == alpha && e.df == df && (e.k == k || (std::isnan(e.k) && std::isnan(k)))) return e.value;
  const double q = q_tukey_ppf(1.0 - alpha, k, df);
  const double t = q / std::sqrt(2.0);
  entries.push_back({alpha, k, df, t * t});
  return t * t;
}

// CritValueTable::build_with_tukey_k (critvals.rs:827-1012).  Row layout per sample size:
// [t_crit_sq_uncorrected, correction_t_crit_sq[0..m), posthoc_t_crit_sq, joint_t_crit_sq, overall_crit]
// `n_predictors` is the FITTED column count (the full P, or p_red for a reduced-model round, batch.rs:2334);
// `tukey_k` (length m, may be null unless correction == TukeyHsd) is the per-target studentized-range k.
int build_crit_rows(int estimator, int correction, double alpha, double posthoc_alpha, uint32_t n_predictors,
                    uint32_t m, const uint32_t* ns, uint32_t n_ns, const double* tukey_k, TukeyMemo* memo,
                    std::vector<double>& out, std::string& err) {
  if (!std::isfinite(alpha) || alpha <= 0.0 || alpha >= 1.0) {
    err = "invalid spec: alpha must lie in (0, 1); got " + std::to_string(alpha);
    return MCP_ERR_INVALID_SPEC;
  }
  const double pa = std::isnan(posthoc_alpha) ? alpha : posthoc_alpha;
  if (!std::isfinite(pa) || pa <= 0.0 || pa >= 1.0) {
    err = "invalid spec: posthoc_alpha must lie in (0, 1); got " + std::to_string(pa);
    return MCP_ERR_INVALID_SPEC;
  }
  const bool student = estimator == MCP_EST_OLS;  // EstimatorSpec::uses_student_t
  const uint32_t stride = m + 4;
  out.assign((size_t)n_ns * stride, 0.0);
  for (uint32_t g = 0; g < n_ns; ++g) {
    double* row = out.data() + (size_t)g * stride;
    const int64_t dfs = (int64_t)ns[g] - (int64_t)n_predictors;
    if (dfs < 1) {
      err = "invalid spec: df = N - P = " + std::to_string(ns[g]) + " - " + std::to_string(n_predictors) + " < 1";
      return MCP_ERR_INVALID_SPEC;
    }
    const double df = (double)dfs;
    auto quant = [&](double pp) { return student ? t_ppf(pp, df) : norm_ppf(pp); };
    const double t_unc = quant(1.0 - alpha / 2.0);
    row[0] = t_unc * t_unc;
    for (uint32_t k = 0; k < m; ++k) {
      double t = t_unc;
      if (correction == MCP_CORR_BONFERRONI) {
        t = quant(1.0 - alpha / (2.0 * m));
      } else if (correction == MCP_CORR_HOLM) {
        const double eff = alpha / (2.0 * (m - k));
        t = eff < 1e-12 ? kInf : quant(1.0 - eff);
      } else if (correction == MCP_CORR_BH) {
        const double eff = double(k + 1) / double(m) * alpha / 2.0;
        t = eff < 1e-12 ? kInf : quant(1.0 - eff);
      } else if (correction == MCP_CORR_TUKEY_HSD) {  // critvals.rs:927-948
        const double kk = tukey_k ? tukey_k[k] : kNaN;
        TukeyMemo local;
        row[1 + k] = (memo ? memo : &local)->threshold(alpha, kk, df);
        continue;
      }
      row[1 + k] = t * t;
    }
    const double t_post = quant(1.0 - pa / 2.0);
    row[1 + m] = t_post * t_post;
    row[2 + m] = (estimator == MCP_EST_MLE && m > 0) ? chi2_ppf(1.0 - alpha, (double)m) : kNaN;
    const double npni = n_predictors > 1 ? double(n_predictors - 1) : 0.0;
    if (npni < 1.0 || estimator == MCP_EST_MLE) row[3 + m] = kInf;
    else if (estimator == MCP_EST_OLS) row[3 + m] = f_ppf(1.0 - alpha, npni, df);
    else row[3 + m] = chi2_ppf(1.0 - alpha, npni);
  }
  return MCP_OK;
}

}  // namespace mcp
