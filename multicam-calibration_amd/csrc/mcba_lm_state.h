// mcba_lm_state.h -- layout of the device-resident Levenberg-Marquardt state and the arguments of the accept / reject
// decision.  No HIP dependency: shared by the kernels (mcba_lm.h) and the host-side unit harness (tests/hostcheck).
#pragma once

// device LM state (doubles): 0 cost  1 lambda  2 nu  3 sel (current slot / linearisation)  4 accepted  5 cost_new  6 pred
// 7 ratio  8 step_norm  9 x_norm  10 dF, then the fields of the device-resident solve (k_solve_cam, mcba_solve.hip):
#define MCBA_LMS 32
#define MCBA_LM_PRED_CAM 11    // d_c^T (lam D_c d_c - g_c) of the camera step waiting in the dc buffer
#define MCBA_LM_DCN2 12        // |d_c|^2
#define MCBA_LM_XCN2 13        // |x_c|^2 at the current point
#define MCBA_LM_SKIP 14        // 1: the reduced solve failed -> this tick only rebuilds the system with more damping
#define MCBA_LM_DONE 15        // 0 running, else the scipy status (1 gtol, 2 ftol, 3 xtol, 4 both): every kernel of a tick returns early
#define MCBA_LM_GINF 16        // first-order optimality at the current point
#define MCBA_LM_NFEV 17        // trial evaluations so far (the host adds the initial one)
#define MCBA_LM_NACC 18        // accepted steps
#define MCBA_LM_PENDING 19     // ftol / xtol verdict of the last decision, applied by the next k_solve_cam
#define MCBA_LM_LAM_USED 20    // damping of the last trial step
#define MCBA_LM_COST_BEFORE 21 // cost before the last trial step
#define MCBA_LM_TICK 22        // ticks that did work
#define MCBA_LM_SOLVE_INFO 23  // 0 ok, 1 reduced system not positive definite / non-finite step, 2 a frame block failed
#define MCBA_LM_REBUILD 24     // 1: the last tick was a damping-only rebuild (no trial)
#define MCBA_LM_CFL 25         // curvature floor of the NEXT linearisations (mcba_math.h: lm_weight): 1 = IRLS, 0.1 = Triggs with a floor
#define MCBA_LM_CFL_SWITCH 26  // > 0: the decision moves MCBA_LM_CFL -- Triggs after an accepted step that gained less than this fraction of the cost, IRLS after a rejected one; 0: fixed
#define MCBA_LM_SEQ 31         // host ring slots only: sequence number of the tick, written last

namespace mcba {
// decide != 0: k_sum_trial / k_decide apply the accept/reject + damping update to the LM state (lms)
// decide == 2: device-resident solve -- pred_cam / dcn2 / xcn2 come from the LM state (k_solve_cam left them there) and
// the ftol / xtol tests of solver.LevenbergMarquardt run on the GPU as well (verdict -> state[MCBA_LM_PENDING])
struct DecideArgs {
  int decide;
  double pred_cam, dcn2, xcn2, lam_min, lam_max;
  double* lms;
  double ftol, xtol;
  // floor of Nielsen's damping factor on an accepted step, lambda *= max(dec_floor, 1 - (2 ratio - 1)^3); 0 = the classical 1/3
  double dec_floor;
};
}  // namespace mcba
