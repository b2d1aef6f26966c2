// mcba_lm.h -- the accept/reject decision of the device-resident LM loop (shared by k_sum_trial / k_decide in
// mcba_kernels.hip and k_solve_cam in mcba_solve.hip).  Compiled into the kernels and, for the decision logic, into the host-side unit harness.
#pragma once
#include "mcba_lm_state.h"
#include "mcba_math.h"

namespace mcba {

// Floor of Nielsen's damping factor: lambda *= max(floor, 1 - (2 ratio - 1)^3) on an accepted step.  Nielsen's 1/3 makes the damping
// the slowest thing in the loop once the steps are good (from the bench's start point: 12 accepted steps just to bring lambda down
// from 210, 24 evaluations to ftol = 1e-4 against 16 with 1/10: profiles/round3/NOTES_round3.md section 5); a caller's 0 means 1/3.
MCBA_HD double lm_dec_floor(double f) { return f > 0.0 ? f : 1.0 / 3.0; }
// damping a speculative Schur reduction assumes for the accepted trial point: Nielsen's factor at its floor
// (the usual case while converging).  Must be the very expression lm_decide evaluates.
MCBA_HD double lm_spec_lambda(double lam, double lam_min, double dec_floor) { return fmax(lam * lm_dec_floor(dec_floor), lam_min); }

// state (lms): 0 cost  1 lambda  2 nu  3 sel (current slot / linearisation)  4 accepted  5 cost_new  6 pred  7 ratio
//        8 step_norm  9 x_norm  10 dF.   pred_cam = d_c^T (lam D_c d_c - g_c), dcn2 = |d_c|^2, xcn2 = |x_c|^2 come from
// whoever solved the camera system (host: DecideArgs; k_solve_cam: the state).  Nielsen's update on acceptance, doubling
// growth on rejection -- identical to solver.LevenbergMarquardt.iterate.
// The state fields the decision reads, fetched in ONE batch (a caller may issue it early, next to its own loads: a
// dependent global round trip costs ~2 us here)
struct LmPre {
  double cost, lam, nu, sel, pred_cam, dcn2, xcn2, nfev, nacc;
};
MCBA_HD void lm_prefetch(const double* lms, LmPre& p) {
  p.cost = lms[0]; p.lam = lms[1]; p.nu = lms[2]; p.sel = lms[3];
  p.pred_cam = lms[MCBA_LM_PRED_CAM]; p.dcn2 = lms[MCBA_LM_DCN2]; p.xcn2 = lms[MCBA_LM_XCN2];
  p.nfev = lms[MCBA_LM_NFEV]; p.nacc = lms[MCBA_LM_NACC];
}
MCBA_HD void lm_decide(const double* trial8, const DecideArgs& da, const LmPre& pre) {
  double* lms = da.lms;
  double cost = pre.cost, lam = pre.lam, nu = pre.nu;
  const bool dev = da.decide == 2;  // camera-step scalars left in the state by k_solve_cam
  const double pred_cam = dev ? pre.pred_cam : da.pred_cam;
  const double dcn2 = dev ? pre.dcn2 : da.dcn2, xcn2 = dev ? pre.xcn2 : da.xcn2;
  const double cost_before = cost, lam_used = lam;
  int sel = static_cast<int>(pre.sel);
  double cost_new = trial8[0];
  double pred = 0.5 * (trial8[1] + pred_cam);
  bool ok = isfinite(cost_new) && pred > 0.0;
  double ratio = ok ? (cost - cost_new) / pred : -1.0;
  double dF = cost - cost_new;
  // round-off guard: near the optimum the last Gauss-Newton corrections change the cost by less than FP64 resolves
  // (|dF| ~ EPS * F * sqrt(m)); such a step is neutral, not bad -- accept it with the damping unchanged (band: MCBA_NEUTRAL_BAND, mcba_math.h)
  bool neutral = isfinite(cost_new) && pred >= 0.0 && fabs(dF) <= MCBA_NEUTRAL_BAND * fabs(cost);
  bool accepted = (ratio > 0.0 && dF >= 0.0) || neutral;
  if (accepted) {
    if (!(ratio > 0.0 && dF >= 0.0)) ratio = 0.5;  // neutral: factor 1 in Nielsen's rule
    double t = 2.0 * ratio - 1.0;
    double fac = fmax(lm_dec_floor(da.dec_floor), 1.0 - t * t * t);
    lam = fmax(lam * fac, da.lam_min);
    nu = 2.0;
    sel ^= 1;
    cost = cost_new;
  } else {
    // (a grey rejection -- the cost rose by less than MCBA_GREY_LEVEL of itself: round-off decides such tests -- doubles the damping
    //  without escalating; mcba_math.h)
    const bool grey = cost_new - cost_before <= MCBA_GREY_LEVEL * cost_before;
    lam = fmin(lam * (grey ? 2.0 : nu), da.lam_max);
    nu = grey ? 2.0 : nu * 2.0;
  }
  const double step_norm = sqrt(trial8[2] + dcn2), x_norm = sqrt(trial8[3] + xcn2);
  lms[0] = cost; lms[1] = lam; lms[2] = nu; lms[3] = sel; lms[4] = accepted ? 1.0 : 0.0;
  lms[5] = cost_new; lms[6] = pred; lms[7] = ratio;
  lms[8] = step_norm; lms[9] = x_norm; lms[10] = dF;
  // curvature model of the next linearisations (mcba_math.h: lm_weight; solver.py: CURV_SWITCH): the majoriser (IRLS) while steps still
  // gain, Triggs' second-order term once an accepted step gained less than the switch fraction of the cost, back after a rejection.
  // Taken here, with the decision, so that every driver (device-resident, host solve, host-driven) changes model at the same step.
  {
    const double sw = lms[MCBA_LM_CFL_SWITCH];
    if (sw > 0.0) {
      // (a rejection at round-off level -- the tail of a converged run -- is no reason to leave Triggs: only a cost that really went up)
      if (!accepted) { if (!(cost_new - cost_before <= MCBA_GREY_LEVEL * cost_before)) lms[MCBA_LM_CFL] = MCBA_CURV_FLOOR_IRLS; }
      else if (neutral || dF < sw * cost_before) lms[MCBA_LM_CFL] = MCBA_CURV_FLOOR_TRIGGS;
    }
  }
  if (dev) {  // termination tests of solver.LevenbergMarquardt._iterate_device, verdict applied by the next k_solve_cam
    const bool ftol_ok = fmax(dF, 0.0) < da.ftol * cost_before && ratio > 0.25;
    const bool xtol_ok = step_norm < da.xtol * (da.xtol + x_norm);
    double status = (ftol_ok && xtol_ok) ? 4.0 : ftol_ok ? 2.0 : xtol_ok ? 3.0 : 0.0;
    if (!accepted && status == 2.0) status = 0.0;  // ftol needs an accepted step
    if (!accepted && lam >= da.lam_max && status == 0.0) status = 3.0;
    lms[MCBA_LM_PENDING] = status;
    lms[MCBA_LM_NFEV] = pre.nfev + 1.0;
    lms[MCBA_LM_NACC] = pre.nacc + (accepted ? 1.0 : 0.0);
    lms[MCBA_LM_LAM_USED] = lam_used;
    lms[MCBA_LM_COST_BEFORE] = cost_before;
    lms[MCBA_LM_REBUILD] = 0.0;
  }
}
MCBA_HD void lm_decide(const double* trial8, const DecideArgs& da) {
  LmPre pre;
  lm_prefetch(da.lms, pre);
  lm_decide(trial8, da, pre);
}
// a tick that only rebuilds the system (the reduced solve failed): no trial, nothing accepted
MCBA_HD void lm_mark_rebuild(double* lms) {
  lms[4] = 0.0;
  lms[MCBA_LM_REBUILD] = 1.0;
}

}  // namespace mcba
