// host_internal.hpp -- declarations shared by the host-side translation units.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mcpower_hip.h"

namespace mcp {

// host_critvals.cpp
double norm_ppf(double p);
double t_ppf(double p, double df);
double chi2_ppf(double p, double k);
double f_ppf(double p, double dfn, double dfd);
int build_crit_rows(int estimator, int correction, double alpha, double posthoc_alpha, uint32_t n_predictors,
                    uint32_t m, const uint32_t* ns, uint32_t n_ns, std::vector<double>& out, std::string& err);

// host_lowering.cpp
struct SimParams;
struct Lowered {
  std::vector<uint8_t> fixed_levels;  // [max_n][n_factors]
  std::vector<double> crit;           // [n_grid][n_tests + 4]
  uint32_t P = 0, n_tests = 0;
  bool need_t3 = false;        // a HighKurtosis marginal can occur: upload the t(3) PPF table
  bool need_beta_eff = false;  // heterogeneity > 0: per-study effects are part of the design record
};
std::vector<double> build_t3_table();
int validate_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, std::string& err);
int lower_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, SimParams& prm, Lowered& low, std::string& err);
void counter_layout(const mcp_spec* spec, mcp_counter_layout* L);

}  // namespace mcp
