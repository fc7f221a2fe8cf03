/* oracle/fit_glm.c -- CPU ORACLE (test infrastructure): GLM (IRLS) arm. Placeholder until the IRLS restatement lands. */
#include "oracle_internal.h"
int mcpo_glm_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                 const double* crit, uint32_t crit_stride, uint8_t* unc, uint8_t* cor, uint8_t* conv, uint8_t* overall,
                 double* stat_row) {
  (void)spec; (void)ws; (void)max_n; (void)ns; (void)n_ns; (void)crit; (void)crit_stride; (void)unc; (void)cor;
  (void)conv; (void)overall; (void)stat_row;
  return MCP_ERR_UNSUPPORTED;
}
