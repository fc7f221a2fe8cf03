/*
 * oracle/rng.c -- CPU ORACLE (test infrastructure): RNG streams and the frozen
 * f32 inverse-normal-CDF kernel.
 *
 * Follows /root/reference/crates/engine-core/src/rng.rs:
 *   pcg_mix64 :36, SimRng :51-128, fill_words :154, fill_uniform_column :183,
 *   fill_normal_column :236, constants :290-325, horner_f32 :330, ln_f32 :343,
 *   norm_inv_cdf_f32 :354.
 * The Philox4x32-10 / SplitMix64-Mix13 / u32_to_unit_f32 primitives live in the
 * third-party crate rand_philox 0.1.0 (Cargo.lock:3929-3932), absent from the
 * tree; they are restated from the published Random123 algorithm (Salmon et al.
 * 2011: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl keys 0x9E3779B9 / 0xBB67AE85)
 * and Stafford's Mix13 finalizer, and pinned by the reference's known-answer
 * vectors in crates/engine-core/tests/rng_byte_identity.rs:18-50.
 *
 * Compile with -ffp-contract=off: every f32 op below must be a separately
 * rounded mul or add (rng.rs:289 "NO fma").
 */
#include <math.h>
#include <string.h>
#include "mcp_oracle.h"

uint64_t mcpo_splitmix64(uint64_t z) {
  /* Mix13 finalizer, no gamma add: splitmix64(0) == 0 (rng_byte_identity.rs:37) */
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}

static inline uint64_t rotl64(uint64_t x, unsigned r) { return (x << r) | (x >> (64 - r)); }

uint64_t mcpo_pcg_mix64(uint64_t base_seed, uint64_t sim_id) {
  /* rng.rs:36-38 */
  return mcpo_splitmix64(base_seed ^ rotl64(sim_id, 32));
}

void mcpo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float mcpo_u32_to_unit_f32(uint32_t w) {
  /* (2*(w>>9)+1) * 2^-24: 23-bit open interval (SURVEY Appendix A;
   * rng_byte_identity.rs:44-50 pins 0 -> 0x33800000, ~0 -> 0x3f7fffff) */
  return (float)(2u * (w >> 9) + 1u) * 5.9604644775390625e-08f;
}

/* ---- frozen f32 inverse normal CDF (rng.rs:290-364) ---------------------- */
static const float NORM_INV_BP = 4.757499993e-01f;
static const float NORM_INV_FLOOR = 2.980232239e-08f;
static const float NORM_INV_LN2 = 6.931471825e-01f;
static const float NORM_INV_LOG2[6] = {
    -2.786813021e+00f, 5.046875954e+00f, -3.492494345e+00f,
    1.593901396e+00f,  -4.048671722e-01f, 4.342890903e-02f,
};
static const float NORM_INV_CENTRAL[11] = {
    2.506664753e+00f,  2.586458445e+00f,  1.238584900e+01f,  -4.216123047e+02f,
    1.463396387e+04f,  -2.779154375e+05f, 3.215264750e+06f,  -2.294391600e+07f,
    9.886485600e+07f,  -2.359132160e+08f, 2.401336800e+08f,
};
static const float NORM_INV_TAIL[10] = {
    -2.150734663e+00f, 2.428994656e+00f,  -7.639000416e-01f, 2.937270701e-01f,
    -8.018484712e-02f, 1.541402005e-02f,  -2.041375730e-03f, 1.774382981e-04f,
    -9.117987247e-06f, 2.100489240e-07f,
};

static float horner_f32(const float* c, int n, float x) {
  /* rng.rs:330-336: acc = acc*x + c, highest coefficient first */
  float acc = c[n - 1];
  for (int k = n - 2; k >= 0; --k) acc = acc * x + c[k]; /* two roundings: built with -ffp-contract=off */
  return acc;
}

static float ln_f32(float t) {
  /* rng.rs:343-348 */
  uint32_t bits;
  memcpy(&bits, &t, 4);
  float e = (float)((int32_t)((bits >> 23) & 0xffu) - 127);
  uint32_t mb = (bits & 0x007fffffu) | 0x3f800000u;
  float m;
  memcpy(&m, &mb, 4);
  return (e + horner_f32(NORM_INV_LOG2, 6, m)) * NORM_INV_LN2;
}

float mcpo_norm_inv_cdf_f32(float u) {
  /* rng.rs:354-364 */
  float v = u - 0.5f;
  float a = fabsf(v);
  if (a <= NORM_INV_BP) {
    return v * horner_f32(NORM_INV_CENTRAL, 11, v * v);
  } else {
    float t = 0.5f - a;
    if (!(t > NORM_INV_FLOOR)) t = NORM_INV_FLOOR; /* f32::max */
    float q = sqrtf(-2.0f * ln_f32(t));
    return copysignf(horner_f32(NORM_INV_TAIL, 10, q), v);
  }
}

/* ---- sequential stream (rng.rs:51-128) ------------------------------------ */
void mcpo_simrng_new(mcpo_simrng* r, uint64_t base_seed, uint64_t sim_id) {
  uint64_t k = mcpo_pcg_mix64(base_seed, sim_id);
  r->key[0] = (uint32_t)k;
  r->key[1] = (uint32_t)(k >> 32);
  r->counter = 0;
  r->buf_pos = 4;
  memset(r->buf, 0, sizeof r->buf);
}

uint32_t mcpo_simrng_next_u32(mcpo_simrng* r) {
  if (r->buf_pos == 4) {
    uint32_t ctr[4] = {(uint32_t)r->counter, (uint32_t)(r->counter >> 32), 0, 0};
    mcpo_philox4x32_10(ctr, r->key, r->buf);
    r->counter += 1;
    r->buf_pos = 0;
  }
  return r->buf[r->buf_pos++];
}

float mcpo_simrng_next_uniform(mcpo_simrng* r) { return mcpo_u32_to_unit_f32(mcpo_simrng_next_u32(r)); }
float mcpo_simrng_next_normal(mcpo_simrng* r) { return mcpo_norm_inv_cdf_f32(mcpo_simrng_next_uniform(r)); }

uint32_t mcpo_simrng_next_categorical(mcpo_simrng* r, const double* probs, uint32_t n) {
  /* rng.rs:112-127 */
  double u = (double)mcpo_simrng_next_uniform(r);
  double acc = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    acc += probs[i];
    if (u < acc) return i;
  }
  return n - 1;
}

/* ---- planar class-addressed fills (rng.rs:130-261) ------------------------ */
void mcpo_fill_words(const uint32_t key[2], uint32_t cls, uint32_t col, uint32_t* out, uint32_t n) {
  uint32_t w[4];
  for (uint32_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = {i >> 2, col, cls, 0};
    mcpo_philox4x32_10(ctr, key, w);
    uint32_t take = n - i < 4 ? n - i : 4;
    for (uint32_t t = 0; t < take; ++t) out[i + t] = w[t];
  }
}

void mcpo_fill_uniform_column(const uint32_t key[2], uint32_t cls, uint32_t col, float* out, uint32_t n) {
  uint32_t w[4];
  for (uint32_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = {i >> 2, col, cls, 0};
    mcpo_philox4x32_10(ctr, key, w);
    uint32_t take = n - i < 4 ? n - i : 4;
    for (uint32_t t = 0; t < take; ++t) out[i + t] = mcpo_u32_to_unit_f32(w[t]);
  }
}

void mcpo_fill_normal_column(const uint32_t key[2], uint32_t cls, uint32_t col, float* out, uint32_t n) {
  /* bit-identical to norm_inv_cdf_f32(u32_to_unit_f32(word)) by the contract at rng.rs:230-231 */
  uint32_t w[4];
  for (uint32_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = {i >> 2, col, cls, 0};
    mcpo_philox4x32_10(ctr, key, w);
    uint32_t take = n - i < 4 ? n - i : 4;
    for (uint32_t t = 0; t < take; ++t) out[i + t] = mcpo_norm_inv_cdf_f32(mcpo_u32_to_unit_f32(w[t]));
  }
}

/* ---- helpers for the golden tests (golden_rng.rs:56-128) ------------------ */
#define STREAM_TAG_SCENARIO 0x5C5C5C5C5C5C5C5CULL /* scenarios.rs:19 */

void mcpo_stream_f32(uint64_t base_seed, uint64_t sim_id, int kind, float* out, uint32_t n) {
  mcpo_simrng r;
  mcpo_simrng_new(&r, base_seed, kind == 2 ? (sim_id ^ STREAM_TAG_SCENARIO) : sim_id);
  for (uint32_t i = 0; i < n; ++i) out[i] = (kind == 0) ? mcpo_simrng_next_normal(&r) : mcpo_simrng_next_uniform(&r);
}

void mcpo_stream_categorical(uint64_t base_seed, uint64_t sim_id, const double* probs, uint32_t n_probs,
                             uint32_t* out, uint32_t n) {
  mcpo_simrng r;
  mcpo_simrng_new(&r, base_seed, sim_id);
  for (uint32_t i = 0; i < n; ++i) out[i] = mcpo_simrng_next_categorical(&r, probs, n_probs);
}

/* ---- distributions.rs:32-63 ----------------------------------------------- */
static double erfc_as(double x) {
  const double A1 = 0.254829592, A2 = -0.284496736, A3 = 1.421413741, A4 = -1.453152027, A5 = 1.061405429;
  const double P = 0.3275911;
  double sign = x < 0.0 ? -1.0 : 1.0;
  double ax = fabs(x);
  double t = 1.0 / (1.0 + P * ax);
  double y = 1.0 - (((((A5 * t + A4) * t) + A3) * t + A2) * t + A1) * t * exp(-ax * ax);
  return 1.0 - sign * y;
}

double mcpo_phi(double z) { return 0.5 * erfc_as(-z * 0.70710678118654752440); }
double mcpo_marginal_uniform(double z, double a, double b) { return a + (b - a) * mcpo_phi(z); }

/* batch probe used by the exhaustive inverse-CDF parity test (rng.rs:539 sweeps all 2^23 reachable uniforms) */
void mcpo_icdf_probe(uint32_t first_word, uint32_t stride, uint32_t n, float* out) {
  for (uint32_t i = 0; i < n; ++i) out[i] = mcpo_norm_inv_cdf_f32(mcpo_u32_to_unit_f32(first_word + i * stride));
}
