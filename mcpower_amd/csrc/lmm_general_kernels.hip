// lmm_general_kernels.hip -- instantiations of the general mixed-model kernel (plain data path); the kernel itself lives in
// lmm_general_impl.hpp.  Split over two translation units so that both data-path flavours compile in parallel.
#include "lmm_general_impl.hpp"

#pragma clang fp contract(off)

namespace mcp {

// LDS bytes of one replicate
size_t lmg_lds_bytes(const SimParams& prm) {
  const size_t W = prm.lmg_fW, LG0 = prm.lmg_levels[0], L0 = prm.lmg_levels[prm.lmg_elim], q0 = prm.lmg_q[0], d = prm.lmg_n_theta;
  const size_t tri = W * (W + 1) / 2, ts = prm.P > 8u ? 16 : 8;  // tail strides (lmg_tail_stride)
  // persistent: gzz | gzw | gww | rval | carry (doubles), ridx (u32, even count), the design (f32)
  const size_t n_pad = ((size_t)prm.max_n + 3u) & ~(size_t)3u;
  const size_t persistent = (L0 * q0 * q0 + L0 * q0 * (size_t)prm.lmg_Wc + tri + LMG_MAX_NZ + LMG_MAX_THETA) * 8 + (((size_t)LMG_MAX_NZ + 1) & ~(size_t)1) * 4 +
                            (size_t)prm.P * (prm.lmg_stream ? (size_t)256 : n_pad) * 4;  // the design, or its 256-row window
  // the shared region: the evaluation's S | B | ld | simplex | fv | tm | misc | tail over the generator's u0 | us | ux | staging
  const size_t eval_bytes = (tri + L0 * q0 * (size_t)prm.lmg_Wc + L0 + (d + 4) * LMG_MAX_THETA + (d + 1) + 27 + 8 + (ts + ts * ts + (ts == 16 ? 12 : 8) * ts + ts * ts + ts + 8)) * 8;
  size_t nlev = 0;
  for (uint32_t t = 1; t < prm.lmg_n_terms; ++t) nlev += (size_t)prm.lmg_levels[t] * prm.lmg_q[t];
  const size_t gen_bytes = (LG0 + LG0 * LMG_MAX_SLOPES + nlev + (size_t)4 * (prm.P + 1) * GW) * 4;
  return persistent + (eval_bytes > gen_bytes ? eval_bytes : gen_bytes) + 64;
}


hipError_t launch_lmm_general(int P, int q0, const SimParams* d_prm, uint32_t n_sims, size_t lds, hipStream_t st) {
  const dim3 block(GW), grid(n_sims);
#define LG(PP, QQ)                                                                                                     \
  if (P == PP && q0 == QQ) {                                                                                           \
    if (lds > 48 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)lmm_general_kernel<PP, QQ, false>,                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return e;                                                                                   \
    }                                                                                                                  \
    hipLaunchKernelGGL((lmm_general_kernel<PP, QQ, false>), grid, block, lds, st, d_prm);                                 \
    return hipGetLastError();                                                                                          \
  }
  LG(2, 1) LG(2, 2) LG(3, 1) LG(3, 2) LG(3, 3) LG(4, 1) LG(4, 2) LG(4, 3) LG(5, 1) LG(5, 2) LG(5, 3) LG(6, 1) LG(6, 2) LG(6, 3) LG(7, 1) LG(7, 2) LG(7, 3) LG(8, 1) LG(8, 2) LG(8, 3) LG(9, 1) LG(10, 1) LG(11, 1) LG(12, 1) LG(9, 2) LG(10, 2) LG(11, 2) LG(12, 2)
#undef LG
  return hipErrorInvalidValue;
}

}  // namespace mcp
