// mcp_device.hpp -- gfx950 device primitives of the simulate path.
//
// The f32 data plane reproduces the reference bit for bit, so this whole
// translation unit is built with -ffp-contract=off and every f32/f64 expression
// on that plane is written as separately rounded mul / add (reference
// crates/engine-core/src/rng.rs:289 "Every op is plain f32 mul/add (NO fma)").
// Fused multiply-add is used only where the reference's own arithmetic is
// out-of-tree f64 linear algebra with a stated tolerance (X'X accumulation,
// Cholesky), via explicit __builtin_fma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mcp {

// ---- seeds (rng.rs:30-38; rand_philox::splitmix64 = Stafford Mix13, no gamma add) ----
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}
__host__ __device__ __forceinline__ uint64_t pcg_mix64(uint64_t base_seed, uint64_t sim_id) {
  return splitmix64(base_seed ^ ((sim_id << 32) | (sim_id >> 32)));  // rotate_left(32)
}

constexpr uint32_t CLASS_XNORM = 1;  // rng.rs:141
constexpr uint32_t CLASS_RESID = 2;  // rng.rs:147
constexpr uint64_t STREAM_TAG_SCENARIO = 0x5C5C5C5C5C5C5C5CULL;  // scenarios.rs:19
constexpr uint64_t STREAM_TAG_HET = 0x4848484848484848ULL;       // scenarios.rs:28

// ---- Philox4x32-10 (Random123; rand_philox 0.1.0) ----
// One 64-bit product per multiplier: v_mad_u64_u32 yields hi and lo together.
struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

// planar word block: rows 4*ctr .. 4*ctr+3 of (class, col)  (rng.rs:149-178)
__device__ __forceinline__ u32x4 planar_words(uint32_t k0, uint32_t k1, uint32_t cls, uint32_t col, uint32_t ctr) {
  return philox4x32_10(ctr, col, cls, 0u, k0, k1);
}
// sequential stream word `q` (0-based draw ordinal) of SimRng (rng.rs:80-90)
__device__ __forceinline__ uint32_t sequential_word(uint32_t k0, uint32_t k1, uint64_t q) {
  const uint64_t c = q >> 2;
  const u32x4 w = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), 0u, 0u, k0, k1);
  const uint32_t l = (uint32_t)q & 3u;
  return l == 0 ? w.x : (l == 1 ? w.y : (l == 2 ? w.z : w.w));
}

// (2*(w>>9)+1) * 2^-24, exact in f32 (rand_philox::u32_to_unit_f32)
__device__ __forceinline__ float u32_to_unit_f32(uint32_t w) {
  return (float)(2u * (w >> 9) + 1u) * 5.9604644775390625e-08f;
}

// ---- frozen f32 inverse normal CDF (rng.rs:290-364) ----
constexpr float NORM_INV_BP = 4.757499993e-01f;
constexpr float NORM_INV_FLOOR = 2.980232239e-08f;
constexpr float NORM_INV_LN2 = 6.931471825e-01f;

__device__ __forceinline__ float icdf_central(float v) {  // v * Pc(v^2), |v| <= BP
  const float r = v * v;
  float acc = 2.401336800e+08f;
  acc = acc * r + -2.359132160e+08f;
  acc = acc * r + 9.886485600e+07f;
  acc = acc * r + -2.294391600e+07f;
  acc = acc * r + 3.215264750e+06f;
  acc = acc * r + -2.779154375e+05f;
  acc = acc * r + 1.463396387e+04f;
  acc = acc * r + -4.216123047e+02f;
  acc = acc * r + 1.238584900e+01f;
  acc = acc * r + 2.586458445e+00f;
  acc = acc * r + 2.506664753e+00f;
  return v * acc;
}

__device__ __forceinline__ float icdf_tail(float v) {  // |v| > BP
  const float a = __builtin_fabsf(v);
  float t = 0.5f - a;
  t = (t > NORM_INV_FLOOR) ? t : NORM_INV_FLOOR;  // f32::max
  // ln_f32 (rng.rs:343-348): exponent + degree-5 mantissa log2 polynomial
  const uint32_t bits = __float_as_uint(t);
  const float e = (float)((int)((bits >> 23) & 0xffu) - 127);
  const float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
  float l = 4.342890903e-02f;
  l = l * m + -4.048671722e-01f;
  l = l * m + 1.593901396e+00f;
  l = l * m + -3.492494345e+00f;
  l = l * m + 5.046875954e+00f;
  l = l * m + -2.786813021e+00f;
  const float ln = (e + l) * NORM_INV_LN2;
  const float q = __builtin_sqrtf(-2.0f * ln);  // IEEE correctly rounded (hipcc default)
  float p = 2.100489240e-07f;
  p = p * q + -9.117987247e-06f;
  p = p * q + 1.774382981e-04f;
  p = p * q + -2.041375730e-03f;
  p = p * q + 1.541402005e-02f;
  p = p * q + -8.018484712e-02f;
  p = p * q + 2.937270701e-01f;
  p = p * q + -7.639000416e-01f;
  p = p * q + 2.428994656e+00f;
  p = p * q + -2.150734663e+00f;
  return __builtin_copysignf(p, v);
}

__device__ __forceinline__ float norm_inv_cdf_f32(float u) {
  const float v = u - 0.5f;
  return (__builtin_fabsf(v) <= NORM_INV_BP) ? icdf_central(v) : icdf_tail(v);
}

// ---- short f64 kernels for the GLM rows (IRLS needs ~1e-15, not the last bit).  Maximum relative error against
// libm (same operations checked in numpy): exp 2.3e-16 for |x| <= 700, log 4.8e-16 on [2^-11, 1].  They pay in
// the logistic row (exp + log1p + reciprocal share their range reductions: 2.27 -> 2.00 ms at C3) but NOT on the
// general data path: swapping ocml's exp / log for them there was measured 7 % slower (13.6 -> 14.6 ms at C5
// realistic), so the marginal transforms keep ocml. ----
__device__ __forceinline__ double fast_rcp(double d) {  // v_rcp_f64 + one Newton step
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
  return y;
}

__device__ __forceinline__ double exp_neg(double a) {  // e^{-a}; any sign of a, |a| <= 700
  const double kf = __builtin_rint(a * 1.4426950408889634);
  double r = __builtin_fma(kf, -6.93147180369123816490e-01, a);  // Cody-Waite: a - k ln2 (hi, lo)
  r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
  const double x = -r;  // |x| <= ln2 / 2
  double p = 1.6059043836821613e-10;  // 1/13!
  p = __builtin_fma(p, x, 2.08767569878681e-09);
  p = __builtin_fma(p, x, 2.505210838544172e-08);
  p = __builtin_fma(p, x, 2.755731922398589e-07);
  p = __builtin_fma(p, x, 2.7557319223985893e-06);
  p = __builtin_fma(p, x, 2.48015873015873e-05);
  p = __builtin_fma(p, x, 0.0001984126984126984);
  p = __builtin_fma(p, x, 0.001388888888888889);
  p = __builtin_fma(p, x, 0.008333333333333333);
  p = __builtin_fma(p, x, 0.041666666666666664);
  p = __builtin_fma(p, x, 0.16666666666666666);
  p = __builtin_fma(p, x, 0.5);
  p = __builtin_fma(p, x, 1.0);
  p = __builtin_fma(p, x, 1.0);
  return ldexp(p, -(int)kf);
}


__device__ __forceinline__ double fast_log(double u) {  // log(u), u > 0 normal
  int e;
  double m = frexp(u, &e);  // m in [0.5, 1)
  const bool small = m < 0.7071067811865476;
  m = small ? m * 2.0 : m;
  const double ef = (double)(small ? e - 1 : e);
  const double s = (m - 1.0) * fast_rcp(m + 1.0);
  const double z = s * s;
  double p = 1.0 / 23.0;
  p = __builtin_fma(p, z, 1.0 / 21.0);
  p = __builtin_fma(p, z, 1.0 / 19.0);
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  p = __builtin_fma(p, z, 1.0);
  return __builtin_fma(ef, 6.93147180369123816490e-01, __builtin_fma(2.0 * s, p, ef * 1.90821492927058770002e-10));
}

}  // namespace mcp
