/* tools/comm_selftest.c -- a C host of the boundary (no Python, no torch): brings up a communicator of ONE rank through
 * include/mcpower_hip.h and checks that a pooled run_batch returns the local counters.  Build + run on a GPU box:
 *   gcc -O1 -I include tools/comm_selftest.c -o tools/_comm_selftest -L mcpower_amd/csrc/_build -lmcpower_hip \
 *       -Wl,-rpath,$PWD/mcpower_amd/csrc/_build && tools/_comm_selftest
 * (multi-rank bring-up: rank 0 writes the id to a file the other ranks read -- see INTEGRATION.md section 3) */
#include <stdio.h>
#include <string.h>
#include "mcpower_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != MCP_OK) { printf("FAIL %s -> %d: %s\n", #call, rc_, mcp_last_error()); return 1; } } while (0)

int main(void) {
  /* C2a of BASELINE.md: k = 3 normals, x1:x2 interaction, p = 5, n = 200 */
  static const double corr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  static const int32_t vt[3] = {MCP_DIST_NORMAL, MCP_DIST_NORMAL, MCP_DIST_NORMAL};
  static const double vp[3] = {0, 0, 0};
  static const double eff[5] = {0.0, 0.18, 0.18, 0.0, 0.10};
  static const uint32_t targets[3] = {1, 2, 4};
  static const uint32_t icomp[2] = {1, 2}, ioff[2] = {0, 2};
  mcp_spec s;
  memset(&s, 0, sizeof s);
  s.abi_version = MCP_ABI_VERSION;
  s.n_non_factor = 3;
  s.correlation = corr; s.var_types = vt; s.var_params = vp; s.n_var_params = 3;
  s.effect_sizes = eff; s.n_effect_sizes = 5;
  s.target_indices = targets; s.n_target_indices = 3;
  s.interaction_components = icomp; s.interaction_offsets = ioff; s.n_interactions = 1;
  s.correction_method = MCP_CORR_NONE; s.alpha = 0.05; s.posthoc_alpha = 0.0 / 0.0;
  s.heteroskedasticity_driver = -1; s.residual_dist = MCP_RESID_NORMAL;
  s.outcome_kind = MCP_OUTCOME_CONTINUOUS; s.estimator = MCP_EST_OLS;
  s.scenario.heteroskedasticity_ratio = 1.0;
  const uint32_t ns[1] = {200};
  const uint64_t seed = 0x9E3779B97F4A7C15ull;

  CHECK(mcp_set_device(0));
  mcp_comm_id id;
  CHECK(mcp_comm_unique_id(&id));
  CHECK(mcp_comm_init(0, 1, &id));
  printf("communicator up: rank %d of %d\n", mcp_comm_rank(), mcp_comm_world());

  uint64_t off, cnt;
  CHECK(mcp_shard_range(100000, mcp_comm_rank(), mcp_comm_world(), &off, &cnt));
  mcp_run_options o;
  memset(&o, 0, sizeof o);
  o.sim_offset = off;
  mcp_batch_result local, pooled;
  CHECK(mcp_run_batch(&s, ns, 1, (uint32_t)cnt, seed, &o, NULL, NULL, &local));
  o.flags = MCP_ALLREDUCE_COUNTERS;
  CHECK(mcp_run_batch(&s, ns, 1, (uint32_t)cnt, seed, &o, NULL, NULL, &pooled));
  const size_t n = (size_t)local.n_sample_sizes * local.layout.stride;
  int same = memcmp(local.counters, pooled.counters, n * sizeof(uint64_t)) == 0;
  printf("pooled == local: %s; k_unc[0..2] = %llu %llu %llu of %llu\n", same ? "yes" : "NO",
         (unsigned long long)pooled.counters[pooled.layout.k_unc], (unsigned long long)pooled.counters[pooled.layout.k_unc + 1],
         (unsigned long long)pooled.counters[pooled.layout.k_unc + 2], (unsigned long long)pooled.counters[pooled.layout.n_sims]);
  mcp_free_result(&local);
  mcp_free_result(&pooled);
  CHECK(mcp_comm_destroy());
  return same ? 0 : 2;
}
