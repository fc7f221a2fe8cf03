// lmm_general_scen_kernels.hip -- instantiations of the general mixed-model kernel (general / scenario data path); the kernel itself lives in
// lmm_general_impl.hpp.  Split over two translation units so that both data-path flavours compile in parallel.
#include "lmm_general_impl.hpp"

#pragma clang fp contract(off)

namespace mcp {

hipError_t launch_lmm_general_scen(int P, int q0, const SimParams* d_prm, uint32_t n_sims, size_t lds, hipStream_t st) {
  const dim3 block(GW), grid(n_sims);
#define LG(PP, QQ)                                                                                                     \
  if (P == PP && q0 == QQ) {                                                                                           \
    if (lds > 48 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)lmm_general_kernel<PP, QQ, true>,                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return e;                                                                                   \
    }                                                                                                                  \
    hipLaunchKernelGGL((lmm_general_kernel<PP, QQ, true>), grid, block, lds, st, d_prm);                                 \
    return hipGetLastError();                                                                                          \
  }
  LG(2, 1) LG(2, 2) LG(3, 1) LG(3, 2) LG(3, 3) LG(4, 1) LG(4, 2) LG(4, 3) LG(5, 1) LG(5, 2) LG(5, 3) LG(6, 1) LG(6, 2) LG(6, 3) LG(7, 1) LG(7, 2) LG(7, 3) LG(8, 1) LG(8, 2) LG(8, 3) LG(9, 1) LG(10, 1) LG(11, 1) LG(12, 1) LG(9, 2) LG(10, 2) LG(11, 2) LG(12, 2)
#undef LG
  return hipErrorInvalidValue;
}

}  // namespace mcp
