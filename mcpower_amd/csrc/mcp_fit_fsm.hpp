// mcp_fit_fsm.hpp -- the optimiser driver of the mixed-model kernels (glmm_kernels.hip, glmm_slopes_kernels.hip,
// lmm_general_impl.hpp) as ONE wave-uniform state machine around ONE evaluation site.
//
// What it runs is what oracle/fit_glmm.c:glmm_fit_model (and its siblings) run as plain calls:
//   bounded Nelder-Mead from simplex row 0 (edge 0.25, stop at df <= 1e-10 and size <= 1e-7)
//   -> stationarity check (mcp_stationary.hpp: central probes, h = 1e-4) -> only if rejected: restarts with edge
//   0.05, 0.01, ... until a run gains <= 1e-8 (at most 4)
//   -> f at the result -> near-zero lower-bounded parameters snapped onto 0 when that costs no objective
//   -> final evaluation (leaves the modes / factors of the accepted point in LDS) -> `on_final`
//   -> optionally lme4's deriv12 stencil (delta = 1e-4) over the free parameters and the lane-0 Schur complement
//      S / 2 onto the trailing d - nth coordinates (mcp_glmm_hessian.hpp's arithmetic, order and all).
// The call-structured form had ~9 call sites of a __noinline__ objective: every call spills the live state around
// it (536-916 B of scratch per lane, 33 GB of scratch writes per 40 000-fit launch in profiles/r02_c_lmm_slope_pmc.csv).
// Here the objective is inlined exactly once; the states hand it the next point.  No function is called: 0 B scratch.
//
// Usage (inside namespace mcp, after the kernel's LDS pointer typedef):
//   FitFsmOut o = simplex_fit_fsm(sx, fv, d, row, lower0, nth, max_eval, lane,
//                                 [&](PT* pt) -> double { clamp pt in place; sync; return objective(pt); },
//                                 [&](double cur, uint32_t n_eval) -> uint32_t { rx covariance ...; return hessian ? FIT_FSM_HESSIAN | theta_free : 0u; });
// sx: simplex block, row stride `row` >= d: rows 0..d vertices, d+1 centroid, d+2 reflection, d+3 expansion /
// probe / stencil point; after a successful Hessian rows 1..d hold H with S / 2 at H[nth + r][nth + c] (lower).
// lower0: bit k = parameter k is bounded below by 0.  Evaluations of the stencil are not counted in n_eval.
// signed_init: the initial simplex of a run steps an unbounded, negative parameter by -edge (oracle/fit_lmm_general.c).
#pragma once

constexpr uint32_t FIT_FSM_HESSIAN = 0x80000000u;

struct FitFsmOut {
  double cur;       // objective at the final point (row 0 of the simplex)
  uint32_t n_eval;  // counted evaluations
  bool hess_ok;     // the stencil ran and every block factored
};

template <class PT, class Eval, class OnFinal>
__device__ __forceinline__ FitFsmOut simplex_fit_fsm(PT* sx, PT* fv, const uint32_t d, const uint32_t row, const uint32_t lower0,
                                                     const uint32_t nth, const uint32_t max_eval, const uint32_t lane,
                                                     Eval&& eval, OnFinal&& on_final, const bool signed_init = false) {
  enum : int { S_NM_BEGIN, S_NM_F0, S_NM_INIT_SETUP, S_NM_INIT_F, S_NM_LOOP, S_NM_FR, S_NM_FE, S_NM_FC, S_NM_SHRINK_SETUP,
               S_NM_SHRINK_F, S_NM_END, S_STAT_SETUP, S_STAT_FP, S_STAT_FM, S_CUR, S_CUR_F, S_SNAP_SETUP, S_SNAP_F, S_FINAL,
               S_FINAL_F, S_HESS_A, S_HESS_DIAG_SETUP, S_HESS_DIAG_F, S_HESS_OFF_B, S_HESS_OFF_SETUP, S_HESS_OFF_F,
               S_HESS_SOLVE, S_DONE };
  PT* cen = sx + (size_t)(d + 1u) * row;
  PT* xr = cen + row;
  PT* xe = xr + row;
  PT* Hm = sx + row;  // H[a][b] at Hm[a * row + b]
  const double inf = __builtin_inf(), dl = 1e-4;
  int st = S_NM_BEGIN, run = 0, restart = 0, hq = 0;
  uint32_t n_eval = 0, vi = 0, lo = 0, hi = 0, snap_k = 0, ha = 0, hb = 0, th_free = 0u, stat_k = 0;
  double nm_step = 0.25, best = 0.0, prev = 0.0, cur = 0.0, f = 0.0;
  double stat_h = 0.0, stat_fp = 0.0;
  bool stat_one = false, hess_ok = false;
  double flo = 0.0, fhi = 0.0, fnh = 0.0, fr = 0.0, th_keep = 0.0, fadd = 0.0, fsub = 0.0, hacc = 0.0;
  bool outside = false, count = true;
  PT* pt = sx;
  // a rejected stationarity check takes the restart loop
  auto stat_reject = [&]() {
    run = 1; restart = 0;
    if (n_eval < max_eval) {
      prev = best;
      nm_step = 0.05;
      st = S_NM_BEGIN;
    } else {
      st = S_CUR;
    }
  };
  for (;;) {
    // ---- advance: consume f (the objective at pt) and set up the next point, until a state asks for an evaluation ----
    bool need = false;
    while (!need && st != S_DONE) {
      switch (st) {
        case S_NM_BEGIN:  // one bounded Nelder-Mead run from simplex row 0 with initial edge nm_step
          pt = sx; count = true; need = true; st = S_NM_F0;
          break;
        case S_NM_F0:
          if (lane == 0u) fv[0] = f;
          vi = 0; st = S_NM_INIT_SETUP;
          break;
        case S_NM_INIT_SETUP:
          if (vi < d) {
            PT* v = sx + (size_t)(vi + 1u) * row;
            // signed_init (general LMM): an unbounded parameter that sits below 0 opens its edge downwards
            const double edge = (signed_init && !((lower0 >> vi) & 1u) && !(sx[vi] >= 0.0)) ? -nm_step : nm_step;
            if (lane < d) v[lane] = sx[lane] + (lane == vi ? edge : 0.0);
            __syncthreads();
            pt = v; count = true; need = true; st = S_NM_INIT_F;
          } else {
            __syncthreads();
            st = S_NM_LOOP;
          }
          break;
        case S_NM_INIT_F:
          if (lane == 0u) fv[vi + 1u] = f;
          ++vi; st = S_NM_INIT_SETUP;
          break;
        case S_NM_LOOP: {
          if (!(n_eval < max_eval)) { st = S_NM_END; break; }
          lo = 0; hi = 0;
          for (uint32_t i = 1; i <= d; ++i) {
            if (fv[i] < fv[lo]) lo = i;
            if (fv[i] > fv[hi]) hi = i;
          }
          uint32_t nh = lo;
          for (uint32_t i = 0; i <= d; ++i)
            if (i != hi && fv[i] > fv[nh]) nh = i;
          double size = 0.0;
          for (uint32_t i = 0; i <= d; ++i)
            for (uint32_t k = 0; k < d; ++k) {
              const double dd = __builtin_fabs(sx[(size_t)i * row + k] - sx[(size_t)lo * row + k]);
              size = dd > size ? dd : size;
            }
          if (fv[hi] < inf && fv[hi] - fv[lo] <= 1e-10 && size <= 1e-7) { st = S_NM_END; break; }
          flo = fv[lo]; fhi = fv[hi]; fnh = fv[nh];
          __syncthreads();
          if (lane < d) {
            double ssum = 0.0;
            for (uint32_t i = 0; i <= d; ++i)
              if (i != hi) ssum += sx[(size_t)i * row + lane];
            const double c = ssum / (double)d;
            cen[lane] = c;
            xr[lane] = c + (c - sx[(size_t)hi * row + lane]);
          }
          __syncthreads();
          pt = xr; count = true; need = true; st = S_NM_FR;
        } break;
        case S_NM_FR: {
          fr = f;
          PT* worst = sx + (size_t)hi * row;
          if (fr < flo) {
            if (lane < d) xe[lane] = cen[lane] + 2.0 * (xr[lane] - cen[lane]);
            __syncthreads();
            pt = xe; count = true; need = true; st = S_NM_FE;
          } else if (fr < fnh) {
            if (lane < d) worst[lane] = xr[lane];
            if (lane == 0u) fv[hi] = fr;
            __syncthreads();
            st = S_NM_LOOP;
          } else {
            outside = fr < fhi;
            if (lane < d) xe[lane] = cen[lane] + 0.5 * ((outside ? xr[lane] : worst[lane]) - cen[lane]);
            __syncthreads();
            pt = xe; count = true; need = true; st = S_NM_FC;
          }
        } break;
        case S_NM_FE: {
          PT* worst = sx + (size_t)hi * row;
          if (lane < d) worst[lane] = f < fr ? xe[lane] : xr[lane];
          if (lane == 0u) fv[hi] = f < fr ? f : fr;
          __syncthreads();
          st = S_NM_LOOP;
        } break;
        case S_NM_FC: {
          PT* worst = sx + (size_t)hi * row;
          if (f < (outside ? fr : fhi)) {
            if (lane < d) worst[lane] = xe[lane];
            if (lane == 0u) fv[hi] = f;
            __syncthreads();
            st = S_NM_LOOP;
          } else {
            vi = 0; st = S_NM_SHRINK_SETUP;
          }
        } break;
        case S_NM_SHRINK_SETUP:
          if (vi <= d && vi == lo) ++vi;
          if (vi <= d) {
            PT* v = sx + (size_t)vi * row;
            if (lane < d) v[lane] = sx[(size_t)lo * row + lane] + 0.5 * (v[lane] - sx[(size_t)lo * row + lane]);
            __syncthreads();
            pt = v; count = true; need = true; st = S_NM_SHRINK_F;
          } else {
            __syncthreads();
            st = S_NM_LOOP;
          }
          break;
        case S_NM_SHRINK_F:
          if (lane == 0u) fv[vi] = f;
          __syncthreads();
          ++vi; st = S_NM_SHRINK_SETUP;
          break;
        case S_NM_END: {  // the best vertex to row 0; then the restart schedule
          uint32_t bl = 0;
          for (uint32_t i = 1; i <= d; ++i)
            if (fv[i] < fv[bl]) bl = i;
          best = fv[bl];
          __syncthreads();
          double best_k = 0.0;
          if (lane < d) best_k = sx[(size_t)bl * row + lane];
          __syncthreads();
          if (lane < d) sx[lane] = best_k;
          __syncthreads();
          if (run == 0) {  // first run: restarts only for a point the stationarity check rejects
            stat_k = 0; st = S_STAT_SETUP;
            break;
          }
          bool again;
          if (!(prev - best > 1e-8)) {
            again = false;
          } else {
            ++restart; again = true;
          }
          if (again && restart < 4 && n_eval < max_eval) {
            prev = best;
            nm_step = 0.01;
            st = S_NM_BEGIN;
          } else {
            st = S_CUR;
          }
        } break;
        case S_STAT_SETUP: {  // probe +h along coordinate stat_k (the single probe of a parameter pinned at 0)
          if (!(best < inf && best > -inf)) { stat_reject(); break; }
          if (stat_k >= d) { st = S_CUR; break; }
          __syncthreads();
          const double pk = sx[stat_k];
          const bool bounded = (lower0 >> stat_k) & 1u;
          stat_h = MCP_STAT_H;
          stat_one = bounded && pk == 0.0;
          if (!stat_one && bounded && pk < stat_h) stat_h = pk;
          if (lane < d) xe[lane] = lane == stat_k ? (stat_one ? stat_h : sx[lane] + stat_h) : sx[lane];
          __syncthreads();
          pt = xe; count = true; need = true; st = S_STAT_FP;
        } break;
        case S_STAT_FP:
          stat_fp = f;
          if (stat_one) {
            if (!(f >= best - MCP_STAT_TOL)) stat_reject();
            else { ++stat_k; st = S_STAT_SETUP; }
            break;
          }
          __syncthreads();
          if (lane < d) xe[lane] = lane == stat_k ? sx[lane] - stat_h : sx[lane];
          __syncthreads();
          pt = xe; count = true; need = true; st = S_STAT_FM;
          break;
        case S_STAT_FM: {
          const double fp = stat_fp, fm = f;
          bool rej = !(fp < inf && fp > -inf && fm < inf && fm > -inf) || fp < best - MCP_STAT_TOL || fm < best - MCP_STAT_TOL;
          if (!rej) {
            const double gk = (fp - fm) / (2.0 * stat_h), Hk = (fp - 2.0 * best + fm) / (stat_h * stat_h);
            rej = Hk > 0.0 ? !(gk * gk <= 2.0 * MCP_STAT_TOL * Hk) : !(__builtin_fabs(fp - fm) <= 2.0 * MCP_STAT_TOL);
          }
          if (rej) stat_reject();
          else { ++stat_k; st = S_STAT_SETUP; }
        } break;
        case S_CUR:
          pt = sx; count = true; need = true; st = S_CUR_F;
          break;
        case S_CUR_F:
          cur = f; snap_k = 0; st = S_SNAP_SETUP;
          break;
        case S_SNAP_SETUP: {  // snap a near-zero bounded parameter onto the boundary when that costs no objective
          bool found = false;
          while (snap_k < d) {
            if ((lower0 >> snap_k) & 1u) {
              th_keep = sx[snap_k];
              if (th_keep != 0.0 && th_keep <= 1e-4) { found = true; break; }
            }
            ++snap_k;
          }
          if (found) {
            __syncthreads();
            if (lane == 0u) sx[snap_k] = 0.0;
            __syncthreads();
            pt = sx; count = true; need = true; st = S_SNAP_F;
          } else {
            st = S_FINAL;
          }
        } break;
        case S_SNAP_F:
          if (f <= cur + 1e-8) cur = f;
          else {
            __syncthreads();
            if (lane == 0u) sx[snap_k] = th_keep;
            __syncthreads();
          }
          ++snap_k; st = S_SNAP_SETUP;
          break;
        case S_FINAL:  // modes and factors of the final point
          pt = sx; count = true; need = true; st = S_FINAL_F;
          break;
        case S_FINAL_F: {
          cur = f;
          const uint32_t want = on_final(cur, n_eval);
          if (want & FIT_FSM_HESSIAN) {
            th_free = want & ~FIT_FSM_HESSIAN;
            ha = 0; st = S_HESS_A;
          } else {
            st = S_DONE;
          }
        } break;
        case S_HESS_A:
          if (ha >= d) { st = S_HESS_SOLVE; break; }
          if (ha < nth && !((th_free >> ha) & 1u)) {  // left out: identity row / column
            __syncthreads();
            if (lane < d) {
              Hm[(size_t)ha * row + lane] = lane == ha ? 1.0 : 0.0;
              Hm[(size_t)lane * row + ha] = lane == ha ? 1.0 : 0.0;
            }
            ++ha;
            break;
          }
          hq = 0; st = S_HESS_DIAG_SETUP;
          break;
        case S_HESS_DIAG_SETUP:
          __syncthreads();
          if (lane < d) xe[lane] = lane == ha ? sx[lane] + (hq ? -dl : dl) : sx[lane];
          __syncthreads();
          pt = xe; count = false; need = true; st = S_HESS_DIAG_F;
          break;
        case S_HESS_DIAG_F:
          if (hq == 0) fadd = f;
          else fsub = f;
          if (++hq < 2) { st = S_HESS_DIAG_SETUP; break; }
          if (lane == 0u) Hm[(size_t)ha * row + ha] = fadd / (dl * dl) - 2.0 * cur / (dl * dl) + fsub / (dl * dl);
          hb = 0; st = S_HESS_OFF_B;
          break;
        case S_HESS_OFF_B:
          while (hb < ha && hb < nth && !((th_free >> hb) & 1u)) ++hb;  // that row / column is already identity
          if (hb >= ha) { ++ha; st = S_HESS_A; break; }
          hq = 0; hacc = 0.0; st = S_HESS_OFF_SETUP;
          break;
        case S_HESS_OFF_SETUP:
          __syncthreads();
          if (lane < d) {
            double v = sx[lane];
            if (lane == hb) v = v + ((hq & 2) ? -dl : dl);
            if (lane == ha) v = v + ((hq & 1) ? -dl : dl);
            xe[lane] = v;
          }
          __syncthreads();
          pt = xe; count = false; need = true; st = S_HESS_OFF_F;
          break;
        case S_HESS_OFF_F:
          hacc = hq == 0 ? f : (hq == 3 ? hacc + f : hacc - f);  // f++ - f+- - f-+ + f--, summed in that order
          if (++hq < 4) { st = S_HESS_OFF_SETUP; break; }
          if (lane == 0u) {
            const double h = hacc / (4.0 * dl * dl);
            Hm[(size_t)ha * row + hb] = h;
            Hm[(size_t)hb * row + ha] = h;
          }
          ++hb; st = S_HESS_OFF_B;
          break;
        case S_HESS_SOLVE: {  // S / 2, S = H_bb - H_bt H_tt^-1 H_tb, by lane 0
          __syncthreads();
          if (lane == 0u) {
            bool okh = true;
            const uint32_t pf = d - nth;
            for (uint32_t a = 0; a < d; ++a)
              for (uint32_t b = 0; b < d; ++b) {
                const double v = Hm[(size_t)a * row + b];
                okh = okh && (v < __builtin_inf()) && (v > -__builtin_inf());
              }
            for (uint32_t j = 0; j < nth; ++j) {  // lower Cholesky of H_tt, in place
              double dj = Hm[(size_t)j * row + j];
              for (uint32_t k = 0; k < j; ++k) dj -= Hm[(size_t)j * row + k] * Hm[(size_t)j * row + k];
              okh = okh && (dj > 0.0);
              const double ljj = __builtin_sqrt(dj);
              Hm[(size_t)j * row + j] = ljj;
              for (uint32_t i = j + 1u; i < nth; ++i) {
                double v = Hm[(size_t)i * row + j];
                for (uint32_t k = 0; k < j; ++k) v -= Hm[(size_t)i * row + k] * Hm[(size_t)j * row + k];
                Hm[(size_t)i * row + j] = v / ljj;
              }
            }
            for (uint32_t r = 0; r < pf; ++r)  // Y = Lt^-1 H_tb, over H_tb
              for (uint32_t i = 0; i < nth; ++i) {
                double v = Hm[(size_t)i * row + nth + r];
                for (uint32_t k = 0; k < i; ++k) v -= Hm[(size_t)i * row + k] * Hm[(size_t)k * row + nth + r];
                Hm[(size_t)i * row + nth + r] = v / Hm[(size_t)i * row + i];
              }
            for (uint32_t r = 0; r < pf; ++r)
              for (uint32_t c = 0; c <= r; ++c) {
                double v = Hm[(size_t)(nth + r) * row + nth + c];
                for (uint32_t k = 0; k < nth; ++k) v -= Hm[(size_t)k * row + nth + r] * Hm[(size_t)k * row + nth + c];
                Hm[(size_t)(nth + r) * row + nth + c] = 0.5 * v;
              }
            xe[0] = okh ? 1.0 : 0.0;
          }
          __syncthreads();
          hess_ok = xe[0] != 0.0;
          __syncthreads();
          st = S_DONE;
        } break;
        default: st = S_DONE; break;
      }
    }
    if (st == S_DONE && !need) break;
    // ---- the ONE evaluation site ----
    n_eval += count ? 1u : 0u;
    f = eval(pt);
  }
  FitFsmOut o;
  o.cur = cur; o.n_eval = n_eval; o.hess_ok = hess_ok;
  return o;
}
