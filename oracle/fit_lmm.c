/* oracle/fit_lmm.c -- CPU ORACLE (test infrastructure): random-intercept LMM (REML) arm. Placeholder until the REML restatement lands. */
#include "oracle_internal.h"
int mcpo_mle_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                 const double* crit, uint32_t crit_stride, uint8_t* unc, uint8_t* cor, uint8_t* conv, uint8_t* bh,
                 uint8_t* joint_unc, uint8_t* joint_cor, double* stat_row) {
  (void)spec; (void)ws; (void)max_n; (void)ns; (void)n_ns; (void)crit; (void)crit_stride; (void)unc; (void)cor;
  (void)conv; (void)bh; (void)joint_unc; (void)joint_cor; (void)stat_row;
  return MCP_ERR_UNSUPPORTED;
}
