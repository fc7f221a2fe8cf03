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
double q_tukey_ppf(double p, double k, double df);
struct TukeyMemo {
  struct Entry { double alpha, k, df, value; };
  std::vector<Entry> entries;
  double threshold(double alpha, double k, double df);  // (q_{1-alpha,k,df} / sqrt 2)^2
};
int build_crit_rows(int estimator, int correction, double alpha, double posthoc_alpha, uint32_t n_predictors,
                    uint32_t m, const uint32_t* ns, uint32_t n_ns, const double* tukey_k, TukeyMemo* memo,
                    std::vector<double>& out, std::string& err);

// host_lowering.cpp
struct SimParams;
struct Lowered {
  std::vector<uint8_t> fixed_levels;  // [max_n][n_factors]
  std::vector<double> crit;           // [n_grid][crit_stride]: n_tests + 4 thresholds (+ post-hoc correction rows)
  std::vector<double> crit_red;       // reduced-design rows, same stride, deduplicated by (N, p_red)
  std::vector<uint16_t> red_row;      // [n_grid][16] -> row of crit_red, 0xffff where the flag combination cannot occur
  uint32_t P = 0, n_tests = 0;
  bool need_t3 = false;        // a HighKurtosis marginal can occur: upload the t(3) PPF table
  bool need_beta_eff = false;  // heterogeneity > 0: per-study effects are part of the design record
};
std::vector<double> build_t3_table();
int validate_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, std::string& err);
int lower_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, SimParams& prm, Lowered& low, std::string& err);
void counter_layout(const mcp_spec* spec, mcp_counter_layout* L);
void tukey_k_per_target(const mcp_spec* spec, std::vector<double>& k);
uint32_t total_posthoc_contrasts(const mcp_spec* spec);
// rows of stride n_tests + 4 + npc at `n_fit` fitted columns (batch.rs:2334-2407 build_crit_tables)
int build_crit_rows_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t n_fit, TukeyMemo* memo,
                         std::vector<double>& out, std::string& err);


// host_comm.cpp: the RCCL communicator of this process (one process per GPU) and the counter all-reduce
void shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t* offset, uint64_t* count);
bool comm_active();
int comm_device();
int comm_rank();
int comm_world();
int comm_probe(std::string& err);
int comm_unique_id(mcp_comm_id* out, int device, std::string& err);
int comm_init(int rank, int world, const mcp_comm_id* id, int device, std::string& err);
int comm_destroy(std::string& err);

}  // namespace mcp
