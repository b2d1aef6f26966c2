// mcba_math.h -- per-observation arithmetic of the bundle-adjustment hot path.
//
// Everything the kernels compute per (camera c, frame f, board point p) lives here as
// small inlined functions, so the same text is compiled into the gfx950 kernels
// (mcba_kernels.hip) and into the host-side unit harness (tests/hostcheck).
//
// Model (reference: multicam_calibration/geometry.py:277-325, bundle_adjustment.py:10-98):
//   X_c = R(rho_c) (R(omega_f) X_o + tau_f) + t_c ;  a = x/z, b = y/z, s = a^2+b^2,
//   d = 1 + k1 s + k2 s^2 ;  u = fx a d + cx, v = fy b d + cy ;  residual = observed - predicted.
//
// "Local" Jacobian.  With P = d(u,v)/dX_c (2x3), R_cf = R_c R_f and A = (X_o x (P R_cf)) row-wise,
//   d(u,v)/d(camera 12)  = [ L_I | A Phi_a + P Phi_b | P ],   d(u,v)/d(pose 6) = [ A Psi_a | P Psi_b ]
// where L_I is the 2x6 intrinsics part and, per (c,f) only (not per point),
//   Phi_a = R_f^T Jr(rho_c), Phi_b = -R_c [tau_f]x Jr(rho_c), Psi_a = Jr(omega_f), Psi_b = R_c,
// Jr = right Jacobian of SO(3).  So the kernels accumulate the 12x12 Gram matrix of the local rows
// L = [L_I | A | P] over the points of one (c,f) -- lane-local, no cross-lane traffic -- and expand it
// ONCE per (c,f) into U_cf (12x12), W_cf (12x6), V_cf (6x6) and the gradient pieces.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MCBA_HD __host__ __device__ __forceinline__
#else
#define MCBA_HD inline
#endif

namespace mcba {

enum Loss { LOSS_LINEAR = 0, LOSS_SOFT_L1 = 1, LOSS_HUBER = 2, LOSS_CAUCHY = 3, LOSS_ARCTAN = 4,
            LOSS_TABLE = 5 };  // the caller's rho, tabulated per observed scalar at the point being linearised (mcba_set_loss_table: least_squares' callable `loss`)

constexpr double MCBA_EPS = 2.220446049250313e-16;
// Resolution of the robust cost as the kernels evaluate it (a sum over up to ~1e7 terms, reduced per wavefront, per workgroup, per
// launch).  A step whose gain is inside MCBA_NEUTRAL_BAND of the cost is NEUTRAL: accepted with the damping unchanged (mcba_lm.h).
// A step that is rejected with a cost increase below MCBA_GREY_LEVEL is a GREY rejection: the accept / reject test is a coin toss
// there (the 9-camera golden evaluates its optimum 55-70 EPS lower than its neighbours), so the damping only doubles -- the escalation
// nu *= 2 is for steps that really went uphill.  Before round 4 five such rejections in a row took the damping from 1e-9 to 0.27 and
// the step-size test ended the run 7e-6 away from the optimum (profiles/round4/NOTES_round4.md section 13).  The same level decides
// whether a rejection sends the curvature model back to IRLS (below).
constexpr double MCBA_NEUTRAL_BAND = 32.0 * MCBA_EPS;
constexpr double MCBA_GREY_LEVEL = 1e-9;
// Curvature weight of the LM's normal equations: scipy's Triggs weight with a floor in units of the IRLS weight rho',
//   w = max(rho' + 2 rho'' f^2, floor * rho'),      floor a RUN-TIME value per linearisation (Sel.cfl, mcba_set_curvature_floor).
// scipy clamps the Triggs weight at EPS (common.py:724-726); for the concave zone of huber / cauchy / arctan (and far outliers of
// soft_l1) that leaves J^T J ~ 0 and Marquardt scaling cannot regularise.  rho' J^T J is the majorising (always descending) quadratic:
//   floor = 1   : w = rho' (rho'' <= 0 for every loss here): plain IRLS -- monotone, robust far from the optimum, linear rate at it;
//   floor = 0.1 : Triggs' second-order term with a safety floor -- fast AT the optimum, but away from it the model UNDER-estimates the
//                 cost along the step wherever residuals are large: steps are rejected, the damping climbs (rounds 1-3 ran on this
//                 alone: 16 evaluations to the reference's default tolerance on the bench problem where IRLS needs 6, and
//                 redescending losses started far from the optimum did not terminate within 400: round 4's randomised sweep,
//                 tests/test_gpu_fuzz.py, profiles/round4/NOTES_round4.md section 13).
// The LM driver (solver.py) starts on IRLS and switches to Triggs when an accepted step gained less than 1 % of the cost, back on a
// rejected step.  The weight only steers the iteration: the stationary point (J^T rho' f = 0) does not depend on it, and the
// materialised Jacobian (k_jacobian -> result.jac) keeps scipy's exact scaling.
constexpr double MCBA_CURV_FLOOR_IRLS = 1.0, MCBA_CURV_FLOOR_TRIGGS = 0.1;

// ---------------------------------------------------------------- fast reciprocal / reciprocal square root
// On the GPU: the hardware seed (v_rcp_f64 / v_rsq_f64: 4.6e-8 / 5.2e-8 relative error, measured on MI355X by
// scripts/micro/rcp_accuracy.hip) plus ONE higher-order correction in FMA form -- r (1 + e + e^2) with e = 1 - x r, and
// y (1 + e/2 + 3 e^2 / 8) with e = 1 - a y^2: cubic convergence, so one step reaches the FP64 rounding level (measured
// 1.1e-16 / 1.4e-16, the same as two Newton steps, with 3 / 5 instructions instead of 4 / 8) -- a quarter of the
// instructions of the IEEE division / sqrt expansions (no denormal / inf fix-ups: arguments here are depths z and
// 1 + (f/f_scale)^2 >= 1).  On the host (unit harness) plain division / sqrt.
MCBA_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, fma(e, e, e), r);
#else
  return 1.0 / x;
#endif
}
MCBA_HD double fast_rsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(a);
  const double e = fma(-a * y, y, 1.0);
  return fma(y, e * fma(0.375, e, 0.5), y);
#else
  return 1.0 / sqrt(a);
#endif
}

// ---------------------------------------------------------------- rotations
// a = sin t/t, b = (1-cos t)/t^2, c = (t-sin t)/t^3 with series below t^2 = 1e-4.
MCBA_HD void rot_coeffs(double th2, double& a, double& b, double& c) {
  if (th2 < 1e-4) {
    a = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0)));
    b = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0))));
    c = (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0))));
  } else {
    // 1 / theta from the reciprocal square root (hardware estimate + one cubic step, full FP64): no square root and none of the
    // three divisions (~25 instructions each) that used to follow; sine and cosine share one range reduction
    const double ith = fast_rsqrt(th2), th = th2 * ith, ith2 = ith * ith;
    double s, co;
    sincos(th, &s, &co);
    a = s * ith;
    b = (1.0 - co) * ith2;
    c = (th - s) * (ith * ith2);
  }
}

// R(r) = I + a [r]x + b [r]x^2  (reference convention, R(0) = I: geometry.py:8-35)
// Jr(r) = I - b [r]x + c [r]x^2 (right Jacobian: R(r + e) ~ R(r) Exp(Jr e))
MCBA_HD void rot_and_jr(const double r[3], double R[9], double Jr[9], const double* abc = nullptr) {
  double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double a, b, c;
  if (abc) { a = abc[0]; b = abc[1]; c = abc[2]; }
  else rot_coeffs(th2, a, b, c);
  double dR = 1.0 - b * th2, dJ = 1.0 - c * th2;
  R[0] = dR + b * r[0] * r[0];
  R[4] = dR + b * r[1] * r[1];
  R[8] = dR + b * r[2] * r[2];
  Jr[0] = dJ + c * r[0] * r[0];
  Jr[4] = dJ + c * r[1] * r[1];
  Jr[8] = dJ + c * r[2] * r[2];
  double r01 = r[0] * r[1], r02 = r[0] * r[2], r12 = r[1] * r[2];
  R[1] = b * r01 - a * r[2];
  R[3] = b * r01 + a * r[2];
  R[2] = b * r02 + a * r[1];
  R[6] = b * r02 - a * r[1];
  R[5] = b * r12 - a * r[0];
  R[7] = b * r12 + a * r[0];
  Jr[1] = c * r01 + b * r[2];
  Jr[3] = c * r01 - b * r[2];
  Jr[2] = c * r02 - b * r[1];
  Jr[6] = c * r02 + b * r[1];
  Jr[5] = c * r12 + b * r[0];
  Jr[7] = c * r12 - b * r[0];
}

// (abc: the three coefficients handed out / taken in, so that a caller needing R now and (R, Jr) later pays for them once)
MCBA_HD void rot_only(const double r[3], double R[9], double* abc = nullptr) {
  double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double a, b, c;
  rot_coeffs(th2, a, b, c);
  if (abc) { abc[0] = a; abc[1] = b; abc[2] = c; }
  double dR = 1.0 - b * th2;
  double r01 = r[0] * r[1], r02 = r[0] * r[2], r12 = r[1] * r[2];
  R[0] = dR + b * r[0] * r[0];
  R[4] = dR + b * r[1] * r[1];
  R[8] = dR + b * r[2] * r[2];
  R[1] = b * r01 - a * r[2];
  R[3] = b * r01 + a * r[2];
  R[2] = b * r02 + a * r[1];
  R[6] = b * r02 - a * r[1];
  R[5] = b * r12 - a * r[0];
  R[7] = b * r12 + a * r[0];
}

// ---------------------------------------------------------------- 3x3 helpers (row-major)
MCBA_HD void mm33(const double* A, const double* B, double* C) {  // C = A B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
MCBA_HD void mtm33(const double* A, const double* B, double* C) {  // C = A^T B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
MCBA_HD void mmt33(const double* A, const double* B, double* C) {  // C = A B^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
MCBA_HD void mv3(const double* A, const double* x, double* y) {  // y = A x
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
MCBA_HD void mtv3(const double* A, const double* x, double* y) {  // y = A^T x
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

// ---------------------------------------------------------------- per-(camera, frame) constants
struct CamConst {   // staged in LDS once per workgroup
  double fx, fy, cx, cy, k1, k2;
  double R[9], t[3], Jr[9];
};
MCBA_HD void make_cam_const(const double cam12[12], CamConst& cc) {
  cc.fx = cam12[0]; cc.fy = cam12[1]; cc.cx = cam12[2]; cc.cy = cam12[3]; cc.k1 = cam12[4]; cc.k2 = cam12[5];
  rot_and_jr(cam12 + 6, cc.R, cc.Jr);
  cc.t[0] = cam12[9]; cc.t[1] = cam12[10]; cc.t[2] = cam12[11];
}

struct PairConst {  // what the point loop needs: X_c = Rcf X_o + tcf
  double Rcf[9], tcf[3];
};
MCBA_HD void make_pair_const(const double* Rc, const double* tc, const double* Rf, const double* tau, PairConst& pc) {
  mm33(Rc, Rf, pc.Rcf);
  mv3(Rc, tau, pc.tcf);
  pc.tcf[0] += tc[0]; pc.tcf[1] += tc[1]; pc.tcf[2] += tc[2];
}

// chain-rule matrices of one (c,f): Phi_a, Phi_b, Psi_a (= Jr_f), Psi_b (= R_c)
struct ChainConst {
  double Pa[9], Pb[9], Sa[9], Sb[9];
};
MCBA_HD void make_chain_const(const double* Rc, const double* Jrc, const double* Rf, const double* Jrf, const double* tau, ChainConst& ch) {
  mtm33(Rf, Jrc, ch.Pa);  // R_f^T Jr_c
  // -R_c [tau]x Jr_c :  [tau]x Jr_c has columns tau x Jr_c[:,j]
  double TJ[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double v0 = Jrc[j], v1 = Jrc[3 + j], v2 = Jrc[6 + j];
    TJ[j] = tau[1] * v2 - tau[2] * v1;
    TJ[3 + j] = tau[2] * v0 - tau[0] * v2;
    TJ[6 + j] = tau[0] * v1 - tau[1] * v0;
  }
  mm33(Rc, TJ, ch.Pb);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    ch.Pb[i] = -ch.Pb[i];
    ch.Sa[i] = Jrf[i];
    ch.Sb[i] = Rc[i];
  }
}

// ---------------------------------------------------------------- robust loss (scipy least_squares.py:160-227, common.py:720-731)
// in : r (residual), inv_fs2 = 1/f_scale^2, fs2 = f_scale^2
// out: rho_half = 0.5 f_scale^2 rho(z)  (this observation's share of the cost)
//      gw = rho'(z)                      (gradient weight:  g = J^T (rho' f))
//      w2 = max(rho' + 2 rho'' f^2, EPS) (scipy's J_scale^2: J~^T J~ = sum w2 j^T j)
// lm_weight() turns (gw, w2) into the curvature weight the LM normal equations use.
MCBA_HD double lm_weight(double gw, double w2, double floor) {
  return fmax(w2, floor * gw);
}
// UNIT: f_scale == 1 (the reference's default), known at compile time: the two multiplications by 1.0 disappear (and 1 + r^2
// becomes one fused multiply-add: results agree with the general code to the last bit or two)
template <int LOSS, bool UNIT = false>
MCBA_HD void loss_weights(double r, double fs2, double inv_fs2, double& rho_half, double& gw, double& w2) {
  double r2 = r * r;
  if (LOSS == LOSS_LINEAR) {
    rho_half = 0.5 * r2; gw = 1.0; w2 = 1.0;
    return;
  }
  if (UNIT) { fs2 = 1.0; inv_fs2 = 1.0; }
  double z = UNIT ? r2 : r2 * inv_fs2;
  double rho0, rho1, rho2;
  if (LOSS == LOSS_SOFT_L1) {
    // rho = 2(sqrt(1+z)-1): rho' = (1+z)^-1/2, rho' + 2 rho'' z = (1+z)^-3/2 -- one rsqrt, no division
    double a = 1.0 + z;
    double it = fast_rsqrt(a);
    double t = a * it;
    rho_half = UNIT ? t - 1.0 : fs2 * (t - 1.0);
    gw = it;
    w2 = fmax(it * it * it, MCBA_EPS);
    return;
  } else if (LOSS == LOSS_HUBER) {
    if (z <= 1.0) { rho0 = z; rho1 = 1.0; rho2 = 0.0; }
    else { double sz = sqrt(z); rho0 = 2.0 * sz - 1.0; rho1 = 1.0 / sz; rho2 = -0.5 * rho1 / z; }
  } else if (LOSS == LOSS_CAUCHY) {
    double it = 1.0 / (1.0 + z);
    rho0 = log1p(z); rho1 = it; rho2 = -it * it;
  } else {
    double it = 1.0 / (1.0 + z * z);
    rho0 = atan(z); rho1 = it; rho2 = -2.0 * z * it * it;
  }
  rho_half = 0.5 * fs2 * rho0;
  gw = rho1;
  double js = rho1 + 2.0 * (rho2 * inv_fs2) * r2;
  w2 = js < MCBA_EPS ? MCBA_EPS : js;
}

// ---------------------------------------------------------------- one point-observation
struct ObsRows {
  double up, vp;          // prediction
  double l0, l1;          // d u/d fx = a d,  d v/d fy = b d
  double l4u, l4v;        // d/d k1: fx a s, fy b s
  double l5u, l5v;        // d/d k2: fx a s^2, fy b s^2
  double Eu[6], Ev[6];    // [A | P] rows
};

struct Intr { double fx, fy, cx, cy, k1, k2; };

MCBA_HD void project_only(const Intr& K, const PairConst& pc, const double Xo[3], double& up, double& vp) {
  double x = fma(pc.Rcf[0], Xo[0], fma(pc.Rcf[1], Xo[1], fma(pc.Rcf[2], Xo[2], pc.tcf[0])));
  double y = fma(pc.Rcf[3], Xo[0], fma(pc.Rcf[4], Xo[1], fma(pc.Rcf[5], Xo[2], pc.tcf[1])));
  double z = fma(pc.Rcf[6], Xo[0], fma(pc.Rcf[7], Xo[1], fma(pc.Rcf[8], Xo[2], pc.tcf[2])));
  double iz = fast_rcp(z);
  double a = x * iz, b = y * iz;
  double s = fma(a, a, b * b);
  double d = fma(s, fma(K.k2, s, K.k1), 1.0);
  up = fma(K.fx * a, d, K.cx);
  vp = fma(K.fy * b, d, K.cy);
}

// Every expression is spelled as an FMA chain (hipcc does not re-associate a*b + c*d + e into two FMAs).
template <bool WITH_INTR>
MCBA_HD void obs_rows_t(const Intr& K, const PairConst& pc, const double Xo[3], ObsRows& o) {
  double x = fma(pc.Rcf[0], Xo[0], fma(pc.Rcf[1], Xo[1], fma(pc.Rcf[2], Xo[2], pc.tcf[0])));
  double y = fma(pc.Rcf[3], Xo[0], fma(pc.Rcf[4], Xo[1], fma(pc.Rcf[5], Xo[2], pc.tcf[1])));
  double z = fma(pc.Rcf[6], Xo[0], fma(pc.Rcf[7], Xo[1], fma(pc.Rcf[8], Xo[2], pc.tcf[2])));
  double iz = fast_rcp(z);
  double a = x * iz, b = y * iz;
  double aa = a * a, bb = b * b;
  double s = aa + bb;
  double d = fma(s, fma(K.k2, s, K.k1), 1.0);
  double dp2 = 2.0 * fma(2.0 * K.k2, s, K.k1);  // 2 d'
  double fa = K.fx * a, fb = K.fy * b;
  o.up = fma(fa, d, K.cx);
  o.vp = fma(fb, d, K.cy);
  if (WITH_INTR) {
    o.l0 = a * d;
    o.l1 = b * d;
    o.l4u = fa * s;
    o.l4v = fb * s;
    o.l5u = o.l4u * s;
    o.l5v = o.l4v * s;
  }
  // D_ab and P = D_ab [[iz,0,-a iz],[0,iz,-b iz]]
  double abdp = (a * b) * dp2;
  double izx = K.fx * iz, izy = K.fy * iz;
  double pu0 = fma(aa, dp2, d) * izx, pu1 = abdp * izx, pu2 = -fma(pu0, a, pu1 * b);
  double pv0 = abdp * izy, pv1 = fma(bb, dp2, d) * izy, pv2 = -fma(pv0, a, pv1 * b);
  o.Eu[3] = pu0; o.Eu[4] = pu1; o.Eu[5] = pu2;
  o.Ev[3] = pv0; o.Ev[4] = pv1; o.Ev[5] = pv2;
  // B = P Rcf (row vectors), A = X_o x B
  double bu0 = fma(pu0, pc.Rcf[0], fma(pu1, pc.Rcf[3], pu2 * pc.Rcf[6]));
  double bu1 = fma(pu0, pc.Rcf[1], fma(pu1, pc.Rcf[4], pu2 * pc.Rcf[7]));
  double bu2 = fma(pu0, pc.Rcf[2], fma(pu1, pc.Rcf[5], pu2 * pc.Rcf[8]));
  double bv0 = fma(pv0, pc.Rcf[0], fma(pv1, pc.Rcf[3], pv2 * pc.Rcf[6]));
  double bv1 = fma(pv0, pc.Rcf[1], fma(pv1, pc.Rcf[4], pv2 * pc.Rcf[7]));
  double bv2 = fma(pv0, pc.Rcf[2], fma(pv1, pc.Rcf[5], pv2 * pc.Rcf[8]));
  o.Eu[0] = fma(Xo[1], bu2, -(Xo[2] * bu1));
  o.Eu[1] = fma(Xo[2], bu0, -(Xo[0] * bu2));
  o.Eu[2] = fma(Xo[0], bu1, -(Xo[1] * bu0));
  o.Ev[0] = fma(Xo[1], bv2, -(Xo[2] * bv1));
  o.Ev[1] = fma(Xo[2], bv0, -(Xo[0] * bv2));
  o.Ev[2] = fma(Xo[0], bv1, -(Xo[1] * bv0));
}
MCBA_HD void obs_rows(const Intr& K, const PairConst& pc, const double Xo[3], ObsRows& o) { obs_rows_t<true>(K, pc, Xo, o); }

// ---- row-sequential form used by k_gram: the quantities both rows share, then ONE row (u or v) at a time, so that
// only one 6-vector [A|P] row (plus its weighted copy) is live while the 87 accumulators are updated.
struct ObsCommon {
  double a, b, s, d, dp2, abdp, izx, izy, fa, fb, up, vp;
  double xr[3];  // Rcf X_o: the point in camera-aligned axes, before the translation
};
// MASKED: lanes without an observation (`ok` false) work on the harmless point (0, 0, 1) instead of their own, so that a
// branch-free caller can weight their rows with 0 without ever meeting inf * 0 (padding frames, cameras that look away).
template <bool MASKED = false>
MCBA_HD void obs_common(const Intr& K, const PairConst& pc, const double Xo[3], ObsCommon& q, bool ok = true) {
  q.xr[0] = fma(pc.Rcf[0], Xo[0], fma(pc.Rcf[1], Xo[1], pc.Rcf[2] * Xo[2]));
  q.xr[1] = fma(pc.Rcf[3], Xo[0], fma(pc.Rcf[4], Xo[1], pc.Rcf[5] * Xo[2]));
  q.xr[2] = fma(pc.Rcf[6], Xo[0], fma(pc.Rcf[7], Xo[1], pc.Rcf[8] * Xo[2]));
  double x = q.xr[0] + pc.tcf[0], y = q.xr[1] + pc.tcf[1], z = q.xr[2] + pc.tcf[2];
  if (MASKED) { x = ok ? x : 0.0; y = ok ? y : 0.0; z = ok ? z : 1.0; }
  double iz = fast_rcp(z);
  q.a = x * iz; q.b = y * iz;
  q.s = fma(q.a, q.a, q.b * q.b);
  q.d = fma(q.s, fma(K.k2, q.s, K.k1), 1.0);
  q.dp2 = 2.0 * fma(2.0 * K.k2, q.s, K.k1);
  q.abdp = (q.a * q.b) * q.dp2;
  q.izx = K.fx * iz; q.izy = K.fy * iz;
  q.fa = K.fx * q.a; q.fb = K.fy * q.b;
  q.up = fma(q.fa, q.d, K.cx);
  q.vp = fma(q.fb, q.d, K.cy);
}
// The same in two halves for a software-pipelined caller: the serial part (rotate, reciprocal) one point ahead -- only
// (a, b, 1/z) travel between the stages --, the cheap polynomial part next to the accumulator updates.
struct ObsLead {
  double a, b, iz, xr[3];
};
// PLANAR: every board point has z = 0 exactly (checked on the host): the third column of the rotation drops out
template <bool MASKED = false, bool PLANAR = false>
MCBA_HD void obs_lead(const PairConst& pc, const double Xo[3], ObsLead& l, bool ok = true) {
  if (PLANAR) {
    l.xr[0] = fma(pc.Rcf[0], Xo[0], pc.Rcf[1] * Xo[1]);
    l.xr[1] = fma(pc.Rcf[3], Xo[0], pc.Rcf[4] * Xo[1]);
    l.xr[2] = fma(pc.Rcf[6], Xo[0], pc.Rcf[7] * Xo[1]);
  } else {
    l.xr[0] = fma(pc.Rcf[0], Xo[0], fma(pc.Rcf[1], Xo[1], pc.Rcf[2] * Xo[2]));
    l.xr[1] = fma(pc.Rcf[3], Xo[0], fma(pc.Rcf[4], Xo[1], pc.Rcf[5] * Xo[2]));
    l.xr[2] = fma(pc.Rcf[6], Xo[0], fma(pc.Rcf[7], Xo[1], pc.Rcf[8] * Xo[2]));
  }
  double x = l.xr[0] + pc.tcf[0], y = l.xr[1] + pc.tcf[1], z = l.xr[2] + pc.tcf[2];
  if (MASKED) { x = ok ? x : 0.0; y = ok ? y : 0.0; z = ok ? z : 1.0; }
  l.iz = fast_rcp(z);
  l.a = x * l.iz; l.b = y * l.iz;
}
MCBA_HD void obs_finish(const Intr& K, const ObsLead& l, ObsCommon& q) {
  q.xr[0] = l.xr[0]; q.xr[1] = l.xr[1]; q.xr[2] = l.xr[2];
  q.a = l.a; q.b = l.b;
  q.s = fma(q.a, q.a, q.b * q.b);
  q.d = fma(q.s, fma(K.k2, q.s, K.k1), 1.0);
  q.dp2 = fma(4.0 * K.k2, q.s, 2.0 * K.k1);  // = 2 (2 k2 s + k1) to the bit; the two products are loop invariants
  q.abdp = (q.a * q.b) * q.dp2;
  q.izx = K.fx * l.iz; q.izy = K.fy * l.iz;
  q.fa = K.fx * q.a; q.fb = K.fy * q.b;
  q.up = fma(q.fa, q.d, K.cx);
  q.vp = fma(q.fb, q.d, K.cy);
}
// ROW 0 = u, 1 = v.  E[0..2] = A~ row, E[3..5] = P row.
// Camera-frame variant used by k_gram: with A = X_o x (Rcf^T p) = ((Rcf X_o) x p)^T Rcf, the rows are accumulated as
// [A~ | P] with A~ = (Rcf X_o) x p -- no product with Rcf per row (9 FMAs and, in k_gram, 18 register-file moves per row
// saved) -- and the constant factor Rcf is folded ONCE per (camera, frame) into the chain matrices (chain_to_cam_rows).
template <int ROW>
MCBA_HD void obs_row_cam(const ObsCommon& q, double E[6]) {
  double p0, p1;
  if (ROW == 0) { p0 = fma(q.a * q.a, q.dp2, q.d) * q.izx; p1 = q.abdp * q.izx; }
  else { p0 = q.abdp * q.izy; p1 = fma(q.b * q.b, q.dp2, q.d) * q.izy; }
  double p2 = -fma(p0, q.a, p1 * q.b);
  E[3] = p0; E[4] = p1; E[5] = p2;
  E[0] = fma(q.xr[1], p2, -(q.xr[2] * p1));
  E[1] = fma(q.xr[2], p0, -(q.xr[0] * p2));
  E[2] = fma(q.xr[0], p1, -(q.xr[1] * p0));
}
// J_rho = A Pa + P Pb = A~ (Rcf Pa) + P Pb,  J_omega = A Sa = A~ (Rcf Sa)
MCBA_HD void chain_to_cam_rows(const double* Rcf, ChainConst& ch) {
  double T[9];
  mm33(Rcf, ch.Pa, T);
#pragma unroll
  for (int i = 0; i < 9; ++i) ch.Pa[i] = T[i];
  mm33(Rcf, ch.Sa, T);
#pragma unroll
  for (int i = 0; i < 9; ++i) ch.Sa[i] = T[i];
}

// ---------------------------------------------------------------- local Gram accumulators of one (c,f)
// The 12x12 Gram matrix of the local rows L = [L_I | A | P] is accumulated by TWO independent roles (two
// wavefronts in k_gram, so each keeps its accumulators in VGPRs with room for a second wave per SIMD):
//   role A: GramA = [A|P]^T [A|P] (21) + its gradient piece (6) + the robust cost
//           -> V_cf, g_f, W rows of (rho, t), U blocks (rho,rho) (rho,t) (t,t), g_c[6:12]
//   role B: GramB = L_I^T L_I (17 structural non-zeros) + L_I^T [A|P] (36) + its gradient piece (6)
//           -> W rows of the intrinsics, U blocks (I,I) (I,rho) (I,t), g_c[0:6]
// Neither needs the other's sums: the expansion splits along the same line.
// Order of `ii`: 0:(0,0) 1:(0,2) 2:(0,4) 3:(0,5) 4:(1,1) 5:(1,3) 6:(1,4) 7:(1,5) 8:(2,2) 9:(2,4) 10:(2,5)
//                11:(3,3) 12:(3,4) 13:(3,5) 14:(4,4) 15:(4,5) 16:(5,5)
// Every accumulation is a chain of single-product "+=" so each becomes ONE v_fma_f64
// (a += x*y + z*w would cost mul + fma + add without -ffast-math reassociation).
struct GramA {
  double ee[21];  // upper triangle of [A|P]^T W [A|P], row-major (i<=j)
  double he[6];   // -[A|P]^T (rho' r)   (residual = obs - pred: minus sign folded in)
  double cost;
};
struct GramB {
  double ii[17];
  double ie[36];  // 6 x 6 : intrinsics x [A|P]
  double hi[6];
};

MCBA_HD void gram_zero(GramA& g) {
#pragma unroll
  for (int i = 0; i < 21; ++i) g.ee[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g.he[i] = 0.0;
  g.cost = 0.0;
}
MCBA_HD void gram_zero(GramB& g) {
#pragma unroll
  for (int i = 0; i < 17; ++i) g.ii[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 36; ++i) g.ie[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g.hi[i] = 0.0;
}

// wu2, wv2: curvature weights (0 for a missing scalar); gu, gv = rho' * residual (0 if missing)
MCBA_HD void gram_add(GramA& g, const ObsRows& o, double wu2, double wv2, double gu, double gv) {
  double Euw[6], Evw[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) { Euw[j] = wu2 * o.Eu[j]; Evw[j] = wv2 * o.Ev[j]; }
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      g.ee[k] += Euw[i] * o.Eu[j];
      g.ee[k] += Evw[i] * o.Ev[j];
      ++k;
    }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    g.he[j] -= o.Eu[j] * gu;
    g.he[j] -= o.Ev[j] * gv;
  }
}

MCBA_HD void gram_add(GramB& g, const ObsRows& o, double wu2, double wv2, double gu, double gv) {
  double Euw[6], Evw[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) { Euw[j] = wu2 * o.Eu[j]; Evw[j] = wv2 * o.Ev[j]; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    g.ie[j] += o.l0 * Euw[j];
    g.ie[6 + j] += o.l1 * Evw[j];
    g.ie[12 + j] += Euw[j];
    g.ie[18 + j] += Evw[j];
    g.ie[24 + j] += o.l4u * Euw[j];
    g.ie[24 + j] += o.l4v * Evw[j];
    g.ie[30 + j] += o.l5u * Euw[j];
    g.ie[30 + j] += o.l5v * Evw[j];
  }
  double w0 = wu2 * o.l0, w1 = wv2 * o.l1, w4u = wu2 * o.l4u, w4v = wv2 * o.l4v, w5u = wu2 * o.l5u, w5v = wv2 * o.l5v;
  g.ii[0] += w0 * o.l0;
  g.ii[1] += w0;
  g.ii[2] += w0 * o.l4u;
  g.ii[3] += w0 * o.l5u;
  g.ii[4] += w1 * o.l1;
  g.ii[5] += w1;
  g.ii[6] += w1 * o.l4v;
  g.ii[7] += w1 * o.l5v;
  g.ii[8] += wu2;
  g.ii[9] += w4u;
  g.ii[10] += w5u;
  g.ii[11] += wv2;
  g.ii[12] += w4v;
  g.ii[13] += w5v;
  g.ii[14] += w4u * o.l4u;
  g.ii[14] += w4v * o.l4v;
  g.ii[15] += w4u * o.l5u;
  g.ii[15] += w4v * o.l5v;
  g.ii[16] += w5u * o.l5u;
  g.ii[16] += w5v * o.l5v;
  g.hi[0] -= o.l0 * gu;
  g.hi[1] -= o.l1 * gv;
  g.hi[2] -= gu;
  g.hi[3] -= gv;
  g.hi[4] -= o.l4u * gu;
  g.hi[4] -= o.l4v * gv;
  g.hi[5] -= o.l5u * gu;
  g.hi[5] -= o.l5v * gv;
}

// One row at a time (ROW 0 = u, 1 = v): w2 = curvature weight, gr = rho' * residual of that scalar.
template <int ROW>
MCBA_HD void gram_add_row(GramA& g, const double E[6], double w2, double gr) {
  double Ew[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) Ew[j] = w2 * E[j];
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) { g.ee[k] += Ew[i] * E[j]; ++k; }
#pragma unroll
  for (int j = 0; j < 6; ++j) g.he[j] -= E[j] * gr;
}
// lf = a d (u) or b d (v);  l4 = fx a s (u) or fy b s (v);  l5 = l4 s
template <int ROW>
MCBA_HD void gram_add_row(GramB& g, const double E[6], double w2, double gr, double lf, double l4, double l5) {
  double Ew[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) Ew[j] = w2 * E[j];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    g.ie[6 * ROW + j] += lf * Ew[j];
    g.ie[12 + 6 * ROW + j] += Ew[j];
    g.ie[24 + j] += l4 * Ew[j];
    g.ie[30 + j] += l5 * Ew[j];
  }
  double wf = w2 * lf, w4 = w2 * l4, w5 = w2 * l5;
  g.ii[4 * ROW + 0] += wf * lf;   // (0,0) | (1,1)
  g.ii[4 * ROW + 1] += wf;        // (0,2) | (1,3)
  g.ii[4 * ROW + 2] += wf * l4;   // (0,4) | (1,4)
  g.ii[4 * ROW + 3] += wf * l5;   // (0,5) | (1,5)
  g.ii[8 + 3 * ROW] += w2;        // (2,2) | (3,3)
  g.ii[9 + 3 * ROW] += w4;        // (2,4) | (3,4)
  g.ii[10 + 3 * ROW] += w5;       // (2,5) | (3,5)
  g.ii[14] += w4 * l4;
  g.ii[15] += w4 * l5;
  g.ii[16] += w5 * l5;
  g.hi[ROW] -= lf * gr;
  g.hi[2 + ROW] -= gr;
  g.hi[4] -= l4 * gr;
  g.hi[5] -= l5 * gr;
}

// ---------------------------------------------------------------- expansion of the local Gram matrix (once per (c,f))
// Outputs are pieces of (for the robust-weighted residual Jacobian):
//   U   : 78 doubles, upper triangle (row-major, i<=j) of J_c^T J_c  (12x12, params fx fy cx cy k1 k2 rho t)
//   gc  : 12        J_c^T f
//   W   : 72        J_c^T J_f  (12x6 row-major; pose params omega, tau)
//   V   : 21        upper triangle of J_f^T J_f
//   gf  : 6         J_f^T f
MCBA_HD int tri12(int i, int j) { return i * 12 - (i * (i - 1)) / 2 + (j - i); }  // i<=j
MCBA_HD int tri6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// role A: writes U[(6..11) x (6..11)], gc[6..11], W rows 6..11, V, gf.  (U, gc, W are full-size arrays.)
MCBA_HD void gram_expand(const GramA& g, const ChainConst& ch, double* U, double* gc, double* W, double* V, double* gf) {
  double HAA[9], HAP[9], HPP[9], HPA[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      HAA[3 * i + j] = g.ee[i <= j ? tri6(i, j) : tri6(j, i)];
      HPP[3 * i + j] = g.ee[i <= j ? tri6(3 + i, 3 + j) : tri6(3 + j, 3 + i)];
      HAP[3 * i + j] = g.ee[tri6(i, 3 + j)];
      HPA[3 * j + i] = HAP[3 * i + j];
    }
  // Q_A = Pa^T HAA + Pb^T HPA ; Q_P = Pa^T HAP + Pb^T HPP   (= J_rho^T L_A, J_rho^T L_P)
  double QA[9], QP[9], T1[9], T2[9];
  mtm33(ch.Pa, HAA, T1); mtm33(ch.Pb, HPA, T2);
#pragma unroll
  for (int i = 0; i < 9; ++i) QA[i] = T1[i] + T2[i];
  mtm33(ch.Pa, HAP, T1); mtm33(ch.Pb, HPP, T2);
#pragma unroll
  for (int i = 0; i < 9; ++i) QP[i] = T1[i] + T2[i];
  double Urr[9];
  mm33(QA, ch.Pa, T1); mm33(QP, ch.Pb, T2);
#pragma unroll
  for (int i = 0; i < 9; ++i) Urr[i] = T1[i] + T2[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (i <= j) {
        U[tri12(6 + i, 6 + j)] = Urr[3 * i + j];
        U[tri12(9 + i, 9 + j)] = HPP[3 * i + j];
      }
      U[tri12(6 + i, 9 + j)] = QP[3 * i + j];
    }
  double Wro[9], Wrt[9], Wto[9], Wtt[9];
  mm33(QA, ch.Sa, Wro); mm33(QP, ch.Sb, Wrt); mm33(HPA, ch.Sa, Wto); mm33(HPP, ch.Sb, Wtt);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      W[6 * (6 + i) + j] = Wro[3 * i + j];
      W[6 * (6 + i) + 3 + j] = Wrt[3 * i + j];
      W[6 * (9 + i) + j] = Wto[3 * i + j];
      W[6 * (9 + i) + 3 + j] = Wtt[3 * i + j];
    }
  double Voo[9], Vot[9], Vtt[9];
  mm33(HAA, ch.Sa, T1); mtm33(ch.Sa, T1, Voo);
  mm33(HAP, ch.Sb, T1); mtm33(ch.Sa, T1, Vot);
  mm33(HPP, ch.Sb, T1); mtm33(ch.Sb, T1, Vtt);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (i <= j) { V[tri6(i, j)] = Voo[3 * i + j]; V[tri6(3 + i, 3 + j)] = Vtt[3 * i + j]; }
      V[tri6(i, 3 + j)] = Vot[3 * i + j];
    }
  double ta[3], tb[3];
  mtv3(ch.Pa, g.he, ta); mtv3(ch.Pb, g.he + 3, tb);
#pragma unroll
  for (int i = 0; i < 3; ++i) { gc[6 + i] = ta[i] + tb[i]; gc[9 + i] = g.he[3 + i]; }
  mtv3(ch.Sa, g.he, gf); mtv3(ch.Sb, g.he + 3, gf + 3);
}

// role B: writes U[(0..5) x (0..11)], gc[0..5], W rows 0..5.
MCBA_HD void gram_expand(const GramB& g, const ChainConst& ch, double* U, double* gc, double* W) {
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) U[tri12(i, j)] = 0.0;
  U[tri12(0, 0)] = g.ii[0];  U[tri12(0, 2)] = g.ii[1];  U[tri12(0, 4)] = g.ii[2];  U[tri12(0, 5)] = g.ii[3];
  U[tri12(1, 1)] = g.ii[4];  U[tri12(1, 3)] = g.ii[5];  U[tri12(1, 4)] = g.ii[6];  U[tri12(1, 5)] = g.ii[7];
  U[tri12(2, 2)] = g.ii[8];  U[tri12(2, 4)] = g.ii[9];  U[tri12(2, 5)] = g.ii[10];
  U[tri12(3, 3)] = g.ii[11]; U[tri12(3, 4)] = g.ii[12]; U[tri12(3, 5)] = g.ii[13];
  U[tri12(4, 4)] = g.ii[14]; U[tri12(4, 5)] = g.ii[15]; U[tri12(5, 5)] = g.ii[16];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double* hia = g.ie + 6 * i;      // H_IA row i
    const double* hip = g.ie + 6 * i + 3;  // H_IP row i
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      U[tri12(i, 6 + j)] = hia[0] * ch.Pa[j] + hia[1] * ch.Pa[3 + j] + hia[2] * ch.Pa[6 + j] + hip[0] * ch.Pb[j] + hip[1] * ch.Pb[3 + j] + hip[2] * ch.Pb[6 + j];
      U[tri12(i, 9 + j)] = hip[j];
      W[6 * i + j] = hia[0] * ch.Sa[j] + hia[1] * ch.Sa[3 + j] + hia[2] * ch.Sa[6 + j];
      W[6 * i + 3 + j] = hip[0] * ch.Sb[j] + hip[1] * ch.Sb[3 + j] + hip[2] * ch.Sb[6 + j];
    }
    gc[i] = g.hi[i];
  }
}

// ---------------------------------------------------------------- materialised Jacobian rows of one point-observation
// Jc[12], Jf[6] for the u row and the v row of the PREDICTION (callers negate / rescale).
MCBA_HD void expand_rows(const ObsRows& o, const ChainConst& ch, double* Jcu, double* Jcv, double* Jfu, double* Jfv) {
  Jcu[0] = o.l0; Jcu[1] = 0.0; Jcu[2] = 1.0; Jcu[3] = 0.0; Jcu[4] = o.l4u; Jcu[5] = o.l5u;
  Jcv[0] = 0.0; Jcv[1] = o.l1; Jcv[2] = 0.0; Jcv[3] = 1.0; Jcv[4] = o.l4v; Jcv[5] = o.l5v;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Jcu[6 + j] = o.Eu[0] * ch.Pa[j] + o.Eu[1] * ch.Pa[3 + j] + o.Eu[2] * ch.Pa[6 + j] + o.Eu[3] * ch.Pb[j] + o.Eu[4] * ch.Pb[3 + j] + o.Eu[5] * ch.Pb[6 + j];
    Jcv[6 + j] = o.Ev[0] * ch.Pa[j] + o.Ev[1] * ch.Pa[3 + j] + o.Ev[2] * ch.Pa[6 + j] + o.Ev[3] * ch.Pb[j] + o.Ev[4] * ch.Pb[3 + j] + o.Ev[5] * ch.Pb[6 + j];
    Jcu[9 + j] = o.Eu[3 + j];
    Jcv[9 + j] = o.Ev[3 + j];
    Jfu[j] = o.Eu[0] * ch.Sa[j] + o.Eu[1] * ch.Sa[3 + j] + o.Eu[2] * ch.Sa[6 + j];
    Jfv[j] = o.Ev[0] * ch.Sa[j] + o.Ev[1] * ch.Sa[3 + j] + o.Ev[2] * ch.Sa[6 + j];
    Jfu[3 + j] = o.Eu[3] * ch.Sb[j] + o.Eu[4] * ch.Sb[3 + j] + o.Eu[5] * ch.Sb[6 + j];
    Jfv[3 + j] = o.Ev[3] * ch.Sb[j] + o.Ev[4] * ch.Sb[3 + j] + o.Ev[5] * ch.Sb[6 + j];
  }
}

// ---------------------------------------------------------------- 6x6 SPD helpers for the frame blocks
// Cholesky of a packed upper-triangle 6x6 (tri6 order) -> L packed lower, row-major by (i>=j): Lp[i(i+1)/2 + j];
// returns false if a pivot is not positive.
// The form the kernels store and consume: the DIAGONAL slots of Lp hold 1 / L_ii (what every forward / backward
// substitution multiplies by), computed with v_rsq_f64 + Newton instead of a square root and fifteen FP64 divisions.
MCBA_HD bool chol6i(const double* Vt, double* Lp) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = Vt[tri6(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lp[i * (i + 1) / 2 + k] * Lp[j * (j + 1) / 2 + k];
      if (i == j) {
        if (!(s > 0.0)) { ok = false; s = 1.0; }
        Lp[i * (i + 1) / 2 + i] = fast_rsqrt(s);
      } else {
        Lp[i * (i + 1) / 2 + j] = s * Lp[j * (j + 1) / 2 + j];
      }
    }
  }
  return ok;
}
// y = L^-1 b   (forward substitution), idiag[i] = 1/L_ii
MCBA_HD void fwd6(const double* Lp, const double* idiag, const double* b, double* y) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= Lp[i * (i + 1) / 2 + k] * y[k];
    y[i] = s * idiag[i];
  }
}
// x = L^-T y  (back substitution)
MCBA_HD void bwd6(const double* Lp, const double* idiag, const double* y, double* x) {
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= Lp[k * (k + 1) / 2 + i] * x[k];
    x[i] = s * idiag[i];
  }
}

}  // namespace mcba
