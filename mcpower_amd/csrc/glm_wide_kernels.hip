// glm_wide_kernels.hip -- the IRLS simulate kernel (glm_impl.hpp) instantiated for WIDE designs, P = 9..16 design columns
// (the reference bench's `glm_wide = glm_spec(15, 0.3)`, crates/engine-core/src/bin/throughput.rs:300-303): one wavefront
// per replicate as for P <= 8, but the fit is irls_fit_wide (glm_impl.hpp): lanes own runs of entries of the normal
// equations rather than rows, the factor lives in LDS, and nothing of size P^2 is held in registers (the row-owning fit
// needs p(p+1)/2 + p + 1 = 153 accumulators per lane at P = 16 and spilled).  Two variants of the superset kernel
// (factors, sparse-factor exclusion, test_formula subsets): logit, and the run-time family (probit / Poisson).  The
// general (scenario) data path stays at P <= 8 (its own limit is 6 continuous predictors).
#define MCP_GLM_WAVES 1  // LDS (design + staging, ~33 KB at n = 200) bounds the occupancy at 4-5 waves per CU anyway
#include "glm_impl.hpp"

namespace mcp {

size_t glm_lds_bytes(int P, uint32_t max_n);

template <bool FACTORS, bool RED, bool LOGIT>
static hipError_t launch_glm_wide_f(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, hipStream_t st) {
  const size_t lds = glm_lds_bytes(P, max_n);
  const dim3 block(WAVE), grid(n_sims);
#define LGW(PP)                                                                                                       \
  case PP: {                                                                                                          \
    if (lds > 48 * 1024) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)glm_sim_kernel<PP, FACTORS, RED, false, LOGIT>,                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return e;                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((glm_sim_kernel<PP, FACTORS, RED, false, LOGIT>), grid, block, lds, st, d_prm);                \
  } break;
  switch (P) {
    LGW(9) LGW(10) LGW(11) LGW(12) LGW(13) LGW(14) LGW(15) LGW(16)
    default: return hipErrorInvalidValue;
  }
#undef LGW
  return hipGetLastError();
}

hipError_t launch_glm_wide(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, bool factors, bool reduced,
                           bool logit, hipStream_t st) {
  // a wide design always lowers with the reduced-fit tables (host_lowering.cpp: `wide`), so the superset kernel serves
  // every spec here; the family stays a template argument (the logit arm without the probit / Poisson expansions)
  (void)factors;
  (void)reduced;
  if (!logit) return launch_glm_wide_f<true, true, false>(P, d_prm, n_sims, max_n, st);
  return launch_glm_wide_f<true, true, true>(P, d_prm, n_sims, max_n, st);
}

}  // namespace mcp
