// ols_scen_kernels.hip -- the general ("scenario") data path of the OLS simulate kernel (ols_sim_impl.hpp, FLAVOR 2):
// per-replicate scenario design, non-normal marginals / residuals, per-study effects, heteroskedastic scaling
// (scenarios.rs:50-186, data_gen.rs:990-1250).  Its own translation unit: the two halves of the OLS kernels build side by side.
// (Built with MachineLICM off -- the ~30 f64 constants of exp / log / Phi then stop sitting in 60 VGPRs for the whole kernel --
// the row kernel fits three waves per SIMD at P = 6, 167 registers, but runs slower: C5 realistic 3.8e8 prefix-fits/s against
// 4.3e8 at two waves with the constants hoisted.  Not kept.)
#include "ols_sim_impl.hpp"

namespace mcp {

hipError_t launch_ols_scen(int mode, int P, const SimParams* d_prm, uint32_t n_sims, uint32_t n_grid, hipStream_t st) {
  if (mode == MODE_FUSED) return launch_sim_pgf<MODE_FUSED, false, 2>(P, d_prm, n_sims, st, n_grid);
  if (mode == MODE_GENERATE) return launch_sim_pgf<MODE_GENERATE, false, 2>(P, d_prm, n_sims, st);
  return hipErrorInvalidValue;
}

}  // namespace mcp
