// The interior-point loop's options, status integers and the two scalar helpers every restatement of the loop shares
// (ipm_core.h: the generic text; wave_ipm.h: the wavefront batch solver).  Free of standard-library includes: the text of
// this file also travels inside the library (ipm_options_src.inc) into the kernels that wave_codegen.h compiles per
// template at run time (hiprtc sees no include path of this tree).
#pragma once
#include <chrono>

#include "exec.h"

namespace dnlp {

struct IpmOptions {
  double tol = 1e-7;                  // reference default (ipopt_nlpif.py:155)
  int max_iter = 3000;
  int mu_strategy = 1;                // 0 monotone, 1 adaptive (reference default :154)
  double mu_init = 0.1;
  double mu_min = 1e-11;
  double mu_max_fact = 1e3;
  double bound_relax_factor = 0.0;    // reference default (:156)
  double bound_push = 1e-2, bound_frac = 1e-2;
  double bound_mult_init_val = 1.0;
  double constr_mult_init_max = 1e3;
  int least_square_init_duals = 1;    // reference default (:159)
  double dual_inf_tol = 1.0, constr_viol_tol = 1e-4, compl_inf_tol = 1e-4;
  double acceptable_tol = 1e-6, acceptable_dual_inf_tol = 1e10, acceptable_constr_viol_tol = 1e-2,
         acceptable_compl_inf_tol = 1e-2, acceptable_obj_change_tol = 1e20;
  int acceptable_iter = 15;
  double nlp_scaling_max_gradient = 100.0;
  int nlp_scaling = 1;
  double kappa_d = 1e-5;
  double max_wall_time = 1e20;
  double diverging_iterates_tol = 1e20;
  double max_hessian_perturbation = 1e20;   // IPOPT's option of the same name and default: beyond it the step computation
                                             // has failed (restoration phase / the mu-strategy ladder) instead of crawling on
  int print_level = 0;
  int max_soc = 4;
  double nlp_inf = 1e19;              // |bound| >= 1e19 means "no bound"
  int max_refine = 10, min_refine = 1;
  int restoration = 1;
  int adaptive_fallback = 1;
  int stall_guard = -1;              // -1 auto: only inside solve() while a rung of the retry ladder is still ahead
                                     // (adaptive_fallback on); 1 always; 0 never (IPOPT has no such rule)
  int lazy_dense_fallback = 0;       // 1: a sparse instance switches to Bunch-Kaufman only in the rungs of the retry
                                     // ladder (the first run handles singular static pivots with delta_c alone)
  int lanczos_inertia_bound = 1;
  int lanczos_min_n = 12000;
  // IPOPT's warm start (warm_start_init_point = yes): start from given primal AND dual values,
  // pushed only slightly into the interior
  // hessian_approximation: 0 exact (the tape's second derivatives), 1 limited-memory (IPOPT's quasi-Newton
  // interior-point mode: BFGS pairs in compact form, no second derivatives evaluated)
  int hessian_approximation = 0;
  int limited_memory_max_history = 6;       // IPOPT default
  int limited_memory_max_skipping = 2;      // IPOPT default: this many rejected updates in a row drop the history
  int warm_start = 0;
  double warm_start_bound_push = 1e-3, warm_start_bound_frac = 1e-3, warm_start_mult_bound_push = 1e-3;
};

DNLP_HD inline double now_sec() {
#if DNLP_DEVICE_PASS
  return 1e-8 * static_cast<double>(wall_clock64());   // constant 100 MHz counter
#else
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
#endif
}

// bound_push / bound_frac projection of one value into (l, u) (WB section 3.6)
DNLP_HD inline double push_into_bounds1(double v, double l, double u, double k1, double k2) {
  const bool hl = l > -kInf, hu = u < kInf;
  if (hl && hu) {
    if (l == u) return l;
    const double pl = fmin(k1 * fmax(1.0, fabs(l)), k2 * (u - l));
    const double pu = fmin(k1 * fmax(1.0, fabs(u)), k2 * (u - l));
    return fmin(fmax(v, l + pl), u - pu);
  }
  if (hl) return fmax(v, l + k1 * fmax(1.0, fabs(l)));
  if (hu) return fmin(v, u - k1 * fmax(1.0, fabs(u)));
  return v;
}

// IPOPT ApplicationReturnStatus values (the integers of ipopt_nlpif.py:31-61)
enum IpmStatus : int {
  Solve_Succeeded = 0, Solved_To_Acceptable_Level = 1, Infeasible_Problem_Detected = 2,
  Search_Direction_Becomes_Too_Small = 3, Diverging_Iterates = 4, User_Requested_Stop = 5,
  Maximum_Iterations_Exceeded = -1,
  Restoration_Failed = -2, Error_In_Step_Computation = -3, Maximum_WallTime_Exceeded = -5,
  Not_Enough_Degrees_Of_Freedom = -10, Invalid_Option = -12, Invalid_Number_Detected = -13,
  Internal_Error = -199
};

}  // namespace dnlp
