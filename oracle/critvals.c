/*
 * oracle/critvals.c -- CPU ORACLE (test infrastructure): once-per-batch
 * critical values in squared space, and the multiplicity corrections.
 *
 * Follows /root/reference/crates/engine-core/src/critvals.rs
 *   norm_ppf :33, betai :91, ln_gamma :112, betacf :138, t_cdf :189, t_ppf :211,
 *   gamma_inc_p :294, chi2_ppf :356, f_pdf :416, f_cdf :431, f_ppf :450,
 *   CritValueTable::build_with_tukey_k :827-1012
 * and /root/reference/crates/engine-core/src/correction.rs
 *   apply_correction :16-92, sort_indices_desc :96-115.
 * Pinned by: critvals.rs:1807-1926 quantile KATs (restated in
 * tests/golden/critvals_kat.json), the `crit` field of every
 * validation/data/*.golden.rds (R qt / qnorm), and scipy.stats in the tests.
 * Tukey-HSD thresholds (q_tukey_ppf :687) are not restated: specs that ask for
 * them are rejected with MCP_ERR_UNSUPPORTED by the oracle.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mcp_oracle.h"

double mcpo_norm_ppf(double p) {
  static const double A[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                              1.38357751867269e2,     -3.066479806614716e+01, 2.506628277459239e+00};
  static const double B[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                              6.680131188771972e+01,  -1.328068155288572e+01};
  static const double C[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                              -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
  static const double D[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                              3.754408661907416e+00};
  const double P_LOW = 0.02425, P_HIGH = 1.0 - 0.02425;
  if (p <= 0.0) return -INFINITY;
  if (p >= 1.0) return INFINITY;
  if (p < P_LOW) {
    double q = sqrt(-2.0 * log(p));
    return (((((C[0] * q + C[1]) * q + C[2]) * q + C[3]) * q + C[4]) * q + C[5]) /
           ((((D[0] * q + D[1]) * q + D[2]) * q + D[3]) * q + 1.0);
  } else if (p <= P_HIGH) {
    double q = p - 0.5, r = q * q;
    return (((((A[0] * r + A[1]) * r + A[2]) * r + A[3]) * r + A[4]) * r + A[5]) * q /
           (((((B[0] * r + B[1]) * r + B[2]) * r + B[3]) * r + B[4]) * r + 1.0);
  } else {
    double q = sqrt(-2.0 * log(1.0 - p));
    return -(((((C[0] * q + C[1]) * q + C[2]) * q + C[3]) * q + C[4]) * q + C[5]) /
           ((((D[0] * q + D[1]) * q + D[2]) * q + D[3]) * q + 1.0);
  }
}

static double ln_gamma_(double x) {
  static const double P[9] = {0.9999999999998099,   676.5203681218851,     -1259.1392167224028,
                              771.3234287776531,    -176.6150291621406,    12.507343278686905,
                              -0.13857109526572012, 9.984369578019572e-6,  1.5056327351493116e-7};
  if (x < 0.5) return log(M_PI / sin(M_PI * x)) - ln_gamma_(1.0 - x);
  x -= 1.0;
  double a = P[0];
  double t = x + 7.0 + 0.5;
  for (int i = 1; i < 9; ++i) a += P[i] / (x + (double)i);
  return 0.5 * log(2.0 * M_PI) + (x + 0.5) * log(t) - t + log(a);
}

static double ln_beta_(double a, double b) { return ln_gamma_(a) + ln_gamma_(b) - ln_gamma_(a + b); }

static double betacf_(double a, double b, double x) {
  const double EPS = 3e-15, FPMIN = 1.0e-300;
  double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < FPMIN) d = FPMIN;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 200; ++m) {
    double mf = (double)m, m2 = 2.0 * mf;
    double aa = mf * (b - mf) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d;
    if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa / c;
    if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    h *= d * c;
    aa = -(a + mf) * (qab + mf) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d;
    if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa / c;
    if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < EPS) return h;
  }
  return h;
}

static double betai_(double a, double b, double x) {
  if (x <= 0.0) return 0.0;
  if (x >= 1.0) return 1.0;
  double bt = exp(a * log(x) + b * log(1.0 - x) - ln_beta_(a, b));
  if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf_(a, b, x) / a;
  return 1.0 - bt * betacf_(b, a, 1.0 - x) / b;
}

static double t_cdf_(double x, double df) {
  if (!isfinite(x)) return x > 0.0 ? 1.0 : 0.0;
  double z = df / (df + x * x);
  double half = 0.5 * betai_(0.5 * df, 0.5, z);
  return x >= 0.0 ? 1.0 - half : half;
}

double mcpo_t_ppf(double p, double df) {
  if (!isfinite(df) || df <= 0.0) return NAN;
  if (p <= 0.0) return -INFINITY;
  if (p >= 1.0) return INFINITY;
  if (p == 0.5) return 0.0;
  int upper = p > 0.5;
  double q = upper ? p : 1.0 - p;
  double x;
  if (df > 200.0) {
    double z = mcpo_norm_ppf(q), z2 = z * z;
    x = z + (z * (z2 + 1.0)) / (4.0 * df) + (z * (5.0 * z2 * z2 + 16.0 * z2 + 3.0)) / (96.0 * df * df);
  } else if (fabs(df - 1.0) < 1e-9) {
    x = fabs(tan(M_PI * (q - 0.5)));
  } else if (fabs(df - 2.0) < 1e-9) {
    double alpha = 4.0 * q * (1.0 - q);
    x = sqrt(2.0 / alpha - 2.0) * (q > 0.5 ? 1.0 : -1.0);
  } else {
    x = mcpo_norm_ppf(q);
  }
  if (df > 1.0e5) return upper ? x : -x;
  double log_norm_const = ln_gamma_(0.5 * (df + 1.0)) - ln_gamma_(0.5 * df) - 0.5 * log(df * M_PI);
  for (int it = 0; it < 80; ++it) {
    double cdf = t_cdf_(x, df);
    double pdf = exp(log_norm_const - 0.5 * (df + 1.0) * log(1.0 + x * x / df));
    if (pdf <= 0.0 || !isfinite(pdf)) break;
    double nx = x - (cdf - q) / pdf;
    if (!isfinite(nx)) break;
    double rel = fabs(nx - x) / (1.0 + fabs(x));
    x = nx;
    if (rel < 1e-14) break;
  }
  return upper ? x : -x;
}

static double gamma_inc_p_(double a, double x) {
  if (x <= 0.0 || a <= 0.0) return 0.0;
  const double EPS = 3e-15, FPMIN = 1.0e-300;
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, term = sum;
    for (int i = 0; i < 200; ++i) {
      ap += 1.0;
      term *= x / ap;
      sum += term;
      if (fabs(term) < fabs(sum) * EPS) return sum * exp(-x + a * log(x) - ln_gamma_(a));
    }
    return sum * exp(-x + a * log(x) - ln_gamma_(a));
  }
  double b = x + 1.0 - a, c = 1.0 / FPMIN, d = 1.0 / b, h = d;
  for (int i = 1; i < 200; ++i) {
    double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < FPMIN) d = FPMIN;
    c = b + an / c;
    if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < EPS) return 1.0 - exp(-x + a * log(x) - ln_gamma_(a)) * h;
  }
  return 1.0 - exp(-x + a * log(x) - ln_gamma_(a)) * h;
}

static double cube_(double v) { return v * v * v; } /* f64::powi(3) */

double mcpo_chi2_ppf(double p, double k) {
  if (!isfinite(k) || k <= 0.0) return NAN;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return INFINITY;
  double z = mcpo_norm_ppf(p), h = 2.0 / (9.0 * k);
  double x = k * cube_(1.0 - h + z * sqrt(h));
  if (!isfinite(x) || x <= 0.0) x = k > 1e-6 ? k : 1e-6;
  double a = 0.5 * k;
  for (int it = 0; it < 80; ++it) {
    double cdf = gamma_inc_p_(a, 0.5 * x);
    double pdf = exp((a - 1.0) * log(0.5 * x) - 0.5 * x - M_LN2 - ln_gamma_(a));
    if (!isfinite(pdf) || pdf <= 0.0) break;
    double nx = x - (cdf - p) / pdf;
    if (!isfinite(nx) || nx <= 0.0) nx = 0.5 * x;
    double rel = fabs(nx - x) / (1.0 + fabs(x));
    x = nx;
    if (rel < 1e-13) break;
  }
  return x;
}

static double f_pdf_(double x, double dfn, double dfd) {
  if (x <= 0.0) return 0.0;
  double a = 0.5 * dfn, b = 0.5 * dfd;
  return exp(a * log(dfn / dfd) + (a - 1.0) * log(x) - (a + b) * log(1.0 + dfn / dfd * x) - ln_beta_(a, b));
}

static double f_cdf_(double x, double dfn, double dfd) {
  if (x <= 0.0) return 0.0;
  if (!isfinite(x)) return 1.0;
  return 1.0 - betai_(0.5 * dfd, 0.5 * dfn, dfd / (dfd + dfn * x));
}

double mcpo_f_ppf(double p, double dfn, double dfd) {
  if (!isfinite(dfn) || !isfinite(dfd) || dfn <= 0.0 || dfd <= 0.0) return NAN;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return INFINITY;
  double z = mcpo_norm_ppf(p), h = 2.0 / (9.0 * dfn);
  double x = dfn * cube_(1.0 - h + z * sqrt(h)) / dfn;
  if (!(x > 1e-6)) x = 1e-6;
  for (int it = 0; it < 80; ++it) {
    double cdf = f_cdf_(x, dfn, dfd), pdf = f_pdf_(x, dfn, dfd);
    if (!isfinite(pdf) || pdf <= 0.0) break;
    double nx = x - (cdf - p) / pdf;
    if (!isfinite(nx) || nx <= 0.0) nx = 0.5 * x;
    double rel = fabs(nx - x) / (1.0 + fabs(x));
    x = nx;
    if (rel < 1e-13) break;
  }
  return x;
}

/* ---- studentized range (Tukey HSD), critvals.rs:486-760 -------------------------------------------
 * 32-node Gauss-Legendre rule on [-1,1] (positive half; nodes/weights of the standard rule). */
static const double GL32_X[16] = {
    0.04830766568773831, 0.1444719615827965, 0.23928736225213706, 0.33186860228212767,
    0.42135127613063533, 0.5068999089322294, 0.5877157572407623, 0.6630442669302152,
    0.7321821187402897, 0.7944837959679424, 0.84936761373257, 0.8963211557660522,
    0.9349060759377397, 0.9647622555875064, 0.9856115115452684, 0.9972638618494816};
static const double GL32_W[16] = {
    0.09654008851472781, 0.09563872007927483, 0.09384439908080457, 0.09117387869576386,
    0.08765209300440391, 0.08331192422694685, 0.07819389578707031, 0.07234579410884845,
    0.06582222277636175, 0.058684093478535704, 0.050998059262376244, 0.042835898022226426,
    0.034273862913021626, 0.025392065309262427, 0.016274394730905965, 0.007018610009469298};

static double norm_pdf_(double z) { return exp(-0.5 * z * z) / sqrt(2.0 * M_PI); }

/* P(range <= w | k iid standard normals) = k * int phi(z) [Phi(z) - Phi(z - w)]^(k-1) dz over [-8, 8],
 * composite GL32 with 4 + (k - 1) panels (critvals.rs:593-612); Phi is the engine's A&S erfc form */
static double range_cdf_(double w, double k) {
  if (w <= 0.0) return 0.0;
  const double lo = -8.0, hi = 8.0;
  uint64_t ki = (uint64_t)k;
  uint64_t panels = 4 + (ki > 0 ? ki - 1 : 0);
  double step = (hi - lo) / (double)panels, half = 0.5 * step, acc = 0.0;
  for (uint64_t pnl = 0; pnl < panels; ++pnl) {
    double mid = lo + ((double)pnl + 0.5) * step;
    for (int i = 0; i < 16; ++i) {
      double dx = half * GL32_X[i], fa, fb, z, inner;
      z = mid + dx; inner = mcpo_phi(z) - mcpo_phi(z - w);
      fa = inner > 0.0 ? norm_pdf_(z) * pow(inner, k - 1.0) : 0.0;
      z = mid - dx; inner = mcpo_phi(z) - mcpo_phi(z - w);
      fb = inner > 0.0 ? norm_pdf_(z) * pow(inner, k - 1.0) : 0.0;
      acc += GL32_W[i] * (fa + fb);
    }
  }
  double v = k * (half * acc);
  return v < 1.0 ? v : 1.0;
}

/* P(W <= q) for k groups and df error df: chi mixing in u = chi2_df/df on a graded mesh (critvals.rs:640-673) */
static double studentized_range_cdf_(double q, double k, double df) {
  if (q <= 0.0) return 0.0;
  if (!isfinite(df) || df > 2.0e4) return range_cdf_(q, k);
  double half_df = 0.5 * df;
  double ln_c = half_df * log(half_df) - ln_gamma_(half_df);
  double u_max = 1.0 + 12.0 * sqrt(2.0 / df);
  int panels = df < 6.0 ? 96 : 24;
  double grade = df < 6.0 ? 3.0 : 2.0;
  double acc = 0.0, a = 0.0;
  for (int pnl = 1; pnl <= panels; ++pnl) {
    double b = u_max * pow((double)pnl / (double)panels, grade);
    double half = 0.5 * (b - a), mid = 0.5 * (b + a);
    for (int i = 0; i < 16; ++i) {
      double dx = half * GL32_X[i], u, ga, gb;
      u = mid + dx;
      ga = u <= 0.0 ? 0.0 : exp(ln_c + (half_df - 1.0) * log(u) - half_df * u) * range_cdf_(q * sqrt(u), k);
      u = mid - dx;
      gb = u <= 0.0 ? 0.0 : exp(ln_c + (half_df - 1.0) * log(u) - half_df * u) * range_cdf_(q * sqrt(u), k);
      acc += half * GL32_W[i] * (ga + gb);
    }
    a = b;
  }
  return acc < 1.0 ? acc : 1.0;
}

/* critvals.rs:687-760: Bonferroni seed, Newton with central differences, bisection fallback */
double mcpo_q_tukey_ppf(double p, double k, double df) {
  if (!isfinite(k) || k <= 1.0 || !isfinite(df) || df < 2.0 || !isfinite(p)) return NAN;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return INFINITY;
  double m = k * (k - 1.0);
  double t_q = mcpo_t_ppf(1.0 - (1.0 - p) / m, df);
  double q = (isfinite(t_q) && t_q > 0.0) ? M_SQRT2 * t_q : 3.0;
  double lo = 0.0, hi = (q > 1.0 ? q : 1.0) * 2.0 + 4.0;
  while (studentized_range_cdf_(hi, k, df) < p && hi < 1.0e3) hi *= 1.5;
  const double h = 1e-4;
  for (int it = 0; it < 60; ++it) {
    double f = studentized_range_cdf_(q, k, df) - p;
    if (fabs(f) < 1e-9) return q;
    if (f < 0.0) lo = q; else hi = q;
    double deriv = (studentized_range_cdf_(q + h, k, df) - studentized_range_cdf_(q - h, k, df)) / (2.0 * h);
    double next = (isfinite(deriv) && deriv > 1e-12) ? q - f / deriv : NAN;
    if (!isfinite(next) || next <= lo || next >= hi) next = 0.5 * (lo + hi);
    if (fabs(next - q) < 1e-10) return next;
    q = next;
  }
  return q;
}

/* batch.rs:2235-2253: studentized-range k of the factor owning kernel column `col`; NaN off the dummy blocks */
static double tukey_k_for_col_(const mcp_spec* spec, uint32_t col) {
  uint32_t base = 1 + spec->n_non_factor;
  if (col < base) return NAN;
  uint32_t cursor = base;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    uint32_t l = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
    uint32_t width = l > 0 ? l - 1 : 0;
    if (col < cursor + width) return (double)l;
    cursor += width;
  }
  return NAN;
}

/* critvals.rs:827-1012; `n_predictors` = total design columns P */
/* `p_total` is the FITTED column count: the full P for the hot-path tables, p_red for a reduced-model round
 * (batch.rs:2334-2407 build_crit_tables is called with either).  `tukey_k` (length m, or NULL) carries the
 * per-target studentized-range k; NULL derives it from the spec's targets (batch.rs:2346-2374). */
int mcpo_build_crit_table_p(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t p_total,
                            const double* tukey_k, double* out) {
  uint32_t m = spec->n_target_indices + spec->n_contrast_pairs;
  double alpha = spec->alpha;
  if (!isfinite(alpha) || alpha <= 0.0 || alpha >= 1.0) return MCP_ERR_INVALID_SPEC;
  double palpha = isnan(spec->posthoc_alpha) ? alpha : spec->posthoc_alpha;
  if (!isfinite(palpha) || palpha <= 0.0 || palpha >= 1.0) return MCP_ERR_INVALID_SPEC;
  int student = spec->estimator == MCP_EST_OLS; /* EstimatorSpec::uses_student_t */
  uint32_t stride = m + 4;
  for (uint32_t g = 0; g < n_ns; ++g) {
    double* row = out + (size_t)g * stride;
    int64_t dfs = (int64_t)ns[g] - (int64_t)p_total;
    if (dfs < 1) return MCP_ERR_INVALID_SPEC;
    double df = (double)dfs;
#define QUANT(pp) (student ? mcpo_t_ppf((pp), df) : mcpo_norm_ppf((pp)))
    double t_unc = QUANT(1.0 - alpha / 2.0);
    row[0] = t_unc * t_unc;
    for (uint32_t k = 0; k < m; ++k) {
      double t;
      switch (spec->correction_method) {
        case MCP_CORR_NONE: t = t_unc; break;
        case MCP_CORR_BONFERRONI: t = QUANT(1.0 - alpha / (2.0 * (double)m)); break;
        case MCP_CORR_HOLM: {
          double eff = alpha / (2.0 * (double)(m - k));
          t = eff < 1e-12 ? INFINITY : QUANT(1.0 - eff);
        } break;
        case MCP_CORR_BH: {
          double eff = (double)(k + 1) / (double)m * alpha / 2.0;
          t = eff < 1e-12 ? INFINITY : QUANT(1.0 - eff);
        } break;
        case MCP_CORR_TUKEY_HSD: { /* critvals.rs:927-948 */
          double kk;
          if (tukey_k) kk = tukey_k[k];
          else if (k < spec->n_target_indices) kk = tukey_k_for_col_(spec, spec->target_indices[k]);
          else kk = tukey_k_for_col_(spec, spec->contrast_pairs[2 * (k - spec->n_target_indices)]);
          t = mcpo_q_tukey_ppf(1.0 - alpha, kk, df) / M_SQRT2;
        } break;
        default: return MCP_ERR_UNSUPPORTED;
      }
      row[1 + k] = t * t;
    }
    double t_post = QUANT(1.0 - palpha / 2.0);
    row[1 + m] = t_post * t_post;
    row[2 + m] = (spec->estimator == MCP_EST_MLE && m > 0) ? mcpo_chi2_ppf(1.0 - alpha, (double)m) : NAN;
    double npni = (double)((int64_t)p_total - 1 > 0 ? (int64_t)p_total - 1 : 0);
    if (npni < 1.0) row[3 + m] = INFINITY;
    else if (spec->estimator == MCP_EST_OLS) row[3 + m] = mcpo_f_ppf(1.0 - alpha, npni, (double)dfs);
    else if (spec->estimator == MCP_EST_GLM) row[3 + m] = mcpo_chi2_ppf(1.0 - alpha, npni);
    else row[3 + m] = INFINITY;
#undef QUANT
  }
  return MCP_OK;
}

int mcpo_build_crit_table(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, double* out) {
  return mcpo_build_crit_table_p(spec, ns, n_ns, 1 + spec->n_non_factor + spec->n_factor_dummies + spec->n_interactions,
                                 NULL, out);
}

/* batch.rs:2262-2313 + :2383-2405: per-N correction thresholds of the post-hoc family, one block per
 * requested factor, concatenated; a block is a CritValueTable with n_targets = C(k,2) at alpha = posthoc_alpha
 * (Tukey: every contrast of the block carries that factor's k).  out: n_ns x npc. */
int mcpo_build_posthoc_corr(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t p_total, double* out) {
  uint32_t npc = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    npc += k * (k - 1) / 2;
  }
  uint32_t c0 = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    uint32_t blk = k * (k - 1) / 2;
    mcp_spec tmp = *spec;
    double pa = isnan(spec->posthoc_alpha) ? spec->alpha : spec->posthoc_alpha;
    tmp.alpha = pa;
    tmp.posthoc_alpha = pa;
    tmp.n_contrast_pairs = 0;
    tmp.n_target_indices = blk;
    double* trow = (double*)malloc(sizeof(double) * n_ns * (blk + 4));
    double* kk = (double*)malloc(sizeof(double) * (blk ? blk : 1));
    for (uint32_t q = 0; q < blk; ++q) kk[q] = (double)k;
    int rc = mcpo_build_crit_table_p(&tmp, ns, n_ns, p_total, kk, trow);
    if (rc == MCP_OK)
      for (uint32_t g = 0; g < n_ns; ++g)
        for (uint32_t q = 0; q < blk; ++q) out[(size_t)g * npc + c0 + q] = trow[(size_t)g * (blk + 4) + 1 + q];
    free(trow);
    free(kk);
    if (rc) return rc;
    c0 += blk;
  }
  return MCP_OK;
}

/* correction.rs:96-115: indices by t^2 descending, NaN last, ties by ascending index */
static void sort_indices_desc_(const double* t_sq, uint32_t t, uint32_t* idx) {
  for (uint32_t i = 0; i < t; ++i) idx[i] = i;
  for (uint32_t i = 1; i < t; ++i) { /* stable insertion sort */
    uint32_t a = idx[i];
    int64_t j = (int64_t)i - 1;
    while (j >= 0) {
      uint32_t b = idx[j];
      double va = t_sq[a], vb = t_sq[b];
      int a_before_b;
      if (isnan(va)) a_before_b = 0;
      else if (isnan(vb)) a_before_b = 1;
      else a_before_b = va > vb; /* equal values keep index order (a > b in index) */
      if (!a_before_b) break;
      idx[j + 1] = b;
      --j;
    }
    idx[j + 1] = a;
  }
}

void mcpo_apply_correction(int method, const double* t_sq, uint32_t t, double crit_unc, const double* crit_sq,
                           uint8_t* out) {
  memset(out, 0, t);
  if (t == 0) return;
  uint32_t order[256];
  if (t > 256) return;
  switch (method) {
    case MCP_CORR_NONE:
      for (uint32_t i = 0; i < t; ++i)
        if (!isnan(t_sq[i]) && t_sq[i] > crit_unc) out[i] = 1;
      break;
    case MCP_CORR_BONFERRONI:
    case MCP_CORR_TUKEY_HSD:
      for (uint32_t i = 0; i < t; ++i)
        if (!isnan(t_sq[i]) && t_sq[i] > crit_sq[i]) out[i] = 1;
      break;
    case MCP_CORR_HOLM:
      sort_indices_desc_(t_sq, t, order);
      for (uint32_t k = 0; k < t; ++k) {
        uint32_t i = order[k];
        if (!isnan(t_sq[i]) && t_sq[i] > crit_sq[k]) out[i] = 1;
        else break;
      }
      break;
    case MCP_CORR_BH: {
      sort_indices_desc_(t_sq, t, order);
      int64_t last = -1;
      for (uint32_t k = 0; k < t; ++k) {
        uint32_t i = order[k];
        if (!isnan(t_sq[i]) && t_sq[i] > crit_sq[k]) last = k;
      }
      for (int64_t k = 0; k <= last; ++k) out[order[k]] = 1;
    } break;
    default: break;
  }
}
