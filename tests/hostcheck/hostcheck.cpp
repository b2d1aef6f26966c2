// Host-side unit harness for csrc/mcba_math.h -- TEST INFRASTRUCTURE ONLY.
// Compiles the device math header with g++ and walks it over a small problem on the CPU so the
// algebra (local Gram matrix + chain-rule expansion, Jacobian rows, robust weights, 6x6 solves), the LM accept /
// reject decision of csrc/mcba_lm.h and (round 6) the per-view arithmetic of calibrate()'s kernels, csrc/mcba_pnp_math.h,
// can be checked against oracle/ba_oracle.py in the GPU-less build container.
// It is never loaded by the product (multicam-calibration_amd/ops.py loads libmcba.so only).
#include "../../multicam-calibration_amd/csrc/mcba_math.h"
#include "../../multicam-calibration_amd/csrc/mcba_lm.h"
#include "../../multicam-calibration_amd/csrc/mcba_pnp_math.h"
#include <cstring>

using namespace mcba;

static double g_curv_floor = MCBA_CURV_FLOOR_IRLS;  // the product's default (hc_set_curvature_floor: 0.1 = Triggs with a floor)

template <int LOSS>
static void weights(double r, bool valid, double fs2, double ifs2, double& cost, double& w2, double& gw) {
  if (!valid) { w2 = 0; gw = 0; return; }
  double rh, g1, ww;
  loss_weights<LOSS>(r, fs2, ifs2, rh, g1, ww);
  cost += rh; w2 = lm_weight(g1, ww, g_curv_floor); gw = g1 * r;
}

static void weights_dyn(int loss, double r, bool valid, double fs2, double ifs2, double& cost, double& w2, double& gw) {
  switch (loss) {
    case LOSS_LINEAR: weights<LOSS_LINEAR>(r, valid, fs2, ifs2, cost, w2, gw); break;
    case LOSS_SOFT_L1: weights<LOSS_SOFT_L1>(r, valid, fs2, ifs2, cost, w2, gw); break;
    case LOSS_HUBER: weights<LOSS_HUBER>(r, valid, fs2, ifs2, cost, w2, gw); break;
    case LOSS_CAUCHY: weights<LOSS_CAUCHY>(r, valid, fs2, ifs2, cost, w2, gw); break;
    default: weights<LOSS_ARCTAN>(r, valid, fs2, ifs2, cost, w2, gw); break;
  }
}

extern "C" {

void hc_set_curvature_floor(double v) { g_curv_floor = v; }

// x: 12C + 6F ; uvs (C,F,N,2) ; outputs U (C,78) gc (C,12) W (C,F,72) V (C,F,21) gf (C,F,6) cost (1)
void hc_normal_eq(int C, int F, int N, const double* uvs, const double* obj, const double* x, int loss, double f_scale,
                  double* U, double* gc, double* W, double* V, double* gf, double* cost) {
  double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  *cost = 0;
  memset(U, 0, sizeof(double) * C * 78);
  memset(gc, 0, sizeof(double) * C * 12);
  for (int c = 0; c < C; ++c) {
    CamConst cc;
    make_cam_const(x + 12 * c, cc);
    Intr K{cc.fx, cc.fy, cc.cx, cc.cy, cc.k1, cc.k2};
    for (int f = 0; f < F; ++f) {
      const double* pose = x + 12 * C + 6 * f;
      double Rf[9], Jrf[9];
      rot_and_jr(pose, Rf, Jrf);
      PairConst pc;
      make_pair_const(cc.R, cc.t, Rf, pose + 3, pc);
      GramA ga;
      GramB gb;
      gram_zero(ga);
      gram_zero(gb);
      for (int p = 0; p < N; ++p) {
        const double* o2 = uvs + (((size_t)c * F + f) * N + p) * 2;
        bool vu = o2[0] == o2[0], vv = o2[1] == o2[1];
        if (!(vu || vv)) continue;
        ObsCommon q;
        obs_common(K, pc, obj + 3 * p, q);
        double wu2, wv2, gu, gv, E[6];
        weights_dyn(loss, o2[0] - q.up, vu, fs2, ifs2, ga.cost, wu2, gu);
        weights_dyn(loss, o2[1] - q.vp, vv, fs2, ifs2, ga.cost, wv2, gv);
        obs_row_cam<0>(q, E);
        gram_add_row<0>(ga, E, wu2, gu);
        gram_add_row<0>(gb, E, wu2, gu, q.a * q.d, q.fa * q.s, q.fa * q.s * q.s);
        obs_row_cam<1>(q, E);
        gram_add_row<1>(ga, E, wv2, gv);
        gram_add_row<1>(gb, E, wv2, gv, q.b * q.d, q.fb * q.s, q.fb * q.s * q.s);
      }
      ChainConst ch;
      make_chain_const(cc.R, cc.Jr, Rf, Jrf, pose + 3, ch);
      chain_to_cam_rows(pc.Rcf, ch);  // as k_gram: camera-frame rows
      double Ul[78], gcl[12];
      gram_expand(ga, ch, Ul, gcl, W + ((size_t)c * F + f) * 72, V + ((size_t)c * F + f) * 21, gf + ((size_t)c * F + f) * 6);
      gram_expand(gb, ch, Ul, gcl, W + ((size_t)c * F + f) * 72);
      for (int i = 0; i < 78; ++i) U[c * 78 + i] += Ul[i];
      for (int i = 0; i < 12; ++i) gc[c * 12 + i] += gcl[i];
      *cost += ga.cost;
    }
  }
}

// prediction (C,F,N,2), Jc (C,F,N,2,12), Jf (C,F,N,2,6) of the PREDICTION
void hc_jac_rows(int C, int F, int N, const double* obj, const double* x, double* pred, double* Jc, double* Jf) {
  for (int c = 0; c < C; ++c) {
    CamConst cc;
    make_cam_const(x + 12 * c, cc);
    Intr K{cc.fx, cc.fy, cc.cx, cc.cy, cc.k1, cc.k2};
    for (int f = 0; f < F; ++f) {
      const double* pose = x + 12 * C + 6 * f;
      double Rf[9], Jrf[9];
      rot_and_jr(pose, Rf, Jrf);
      PairConst pc;
      make_pair_const(cc.R, cc.t, Rf, pose + 3, pc);
      ChainConst ch;
      make_chain_const(cc.R, cc.Jr, Rf, Jrf, pose + 3, ch);
      for (int p = 0; p < N; ++p) {
        size_t i = ((size_t)c * F + f) * N + p;
        ObsRows o;
        obs_rows(K, pc, obj + 3 * p, o);
        pred[2 * i] = o.up; pred[2 * i + 1] = o.vp;
        double up2, vp2;
        project_only(K, pc, obj + 3 * p, up2, vp2);
        if (up2 != o.up || vp2 != o.vp) pred[2 * i] = NAN;  // the two code paths must agree bit for bit
        expand_rows(o, ch, Jc + i * 24, Jc + i * 24 + 12, Jf + i * 12, Jf + i * 12 + 6);
      }
    }
  }
}

// rotation constants
void hc_rot(const double* r, double* R, double* Jr) { rot_and_jr(r, R, Jr); }

// Cholesky solve of a packed 6x6: x = V^-1 b ; returns 1 if positive definite
int hc_chol_solve(const double* Vt, const double* b, double* x) {
  double Lp[21], id[6], y[6];
  bool ok = chol6i(Vt, Lp);  // the form the kernels use: diagonal slots hold 1 / L_ii
  for (int i = 0; i < 6; ++i) id[i] = Lp[i * (i + 1) / 2 + i];
  fwd6(Lp, id, b, y);
  bwd6(Lp, id, y, x);
  return ok ? 1 : 0;
}

// the device loop's accept / reject + damping update + ftol / xtol verdict (csrc/mcba_lm.h) on a 32-double LM state
void hc_lm_decide(double* lms, const double* trial8, double lam_min, double lam_max, double ftol, double xtol, double dec_floor) {
  using std::isfinite;
  lm_decide(trial8, DecideArgs{2, 0.0, 0.0, 0.0, lam_min, lam_max, lms, ftol, xtol, dec_floor});
}

// ---- calibrate()'s per-view arithmetic (csrc/mcba_pnp_math.h, the text k_pnp is made of), one view at a time, as one GPU lane runs it
// (the kernel's wave-uniform loops "until no lane is left" become plain loops here)
static void board_norm(const double* obj, int N, double* bn) {   // mcba_api.hip: board_normalisation
  bn[0] = bn[1] = 0.0; bn[2] = 1.0;
  for (int p = 0; p < N; ++p) { bn[0] += obj[3 * p]; bn[1] += obj[3 * p + 1]; }
  bn[0] /= N; bn[1] /= N;
  double ms = 0.0;
  for (int p = 0; p < N; ++p) ms += (obj[3 * p] - bn[0]) * (obj[3 * p] - bn[0]) + (obj[3 * p + 1] - bn[1]) * (obj[3 * p + 1] - bn[1]);
  if (ms > 0.0) bn[2] = sqrt(2.0) / sqrt(ms / N);
}
// uv (N,2) detections of one view; intr9 NULL: the homography of the pixel coordinates (cv2.calibrateCamera's start), else of the undistorted
// normalised ones (cv2.solvePnP's).  H (9) out; returns 1 if the view is complete and H finite.
int hc_view_homography(int N, const double* uv, const double* obj, const double* intr9, int und_iters, double* H) {
  Cam9 cam{1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (intr9) cam = load_cam9(intr9);
  const double ifx = 1.0 / cam.fx, ify = 1.0 / cam.fy;
  auto image_point = [&](int p, double& x, double& y, bool& present) {
    const double u = uv[2 * p], v = uv[2 * p + 1];
    present = u == u && v == v;
    if (intr9) undistort_norm(u, v, cam, ifx, ify, und_iters, x, y);
    else { x = u; y = v; }
  };
  double bn[3];
  board_norm(obj, N, bn);
  bool complete;
  double mx, my, ss;
  view_normalisation(image_point, N, true, mcba::WholeView{}, complete, mx, my, ss);
  DltFactor dlt;
  view_dlt_factor(image_point, obj, N, bn[0], bn[1], bn[2], complete, mx, my, ss, mcba::WholeView{}, dlt);
  double h[9];
  dlt_start_vector(h);
  for (int it = 0; it < 60; ++it) {
    const double diff = dlt_inverse_iteration(dlt, h);
    if (!(complete && !(diff <= 4e-16))) break;
  }
  homography_denormalise(h, bn[0], bn[1], bn[2], mx, my, ss, H);
  bool ok = complete;
  for (int i = 0; i < 9; ++i) ok = ok && pnp_finite(H[i]);
  return ok ? 1 : 0;
}
// cv2.solvePnP's job for one view: start (6) = the pose from the homography, pose (6) = after at most max_evals linearisations; returns the
// number of linearisations, 0 if the view is incomplete, -1 if no pose came out
int hc_view_pose(int N, const double* uv, const double* obj, const double* intr9, int und_iters, int max_evals, double* start, double* pose, double* cost) {
  double H[9];
  const int complete = hc_view_homography(N, uv, obj, intr9, und_iters, H);
  pose_from_homography(H, start);
  const Cam9 cam = load_cam9(intr9);
  ViewLM lm;
  view_lm_init(lm, start, complete != 0);
  auto observation = [&](int p, double& u, double& v) { u = uv[2 * p]; v = uv[2 * p + 1]; };
  for (int it = 0; it < max_evals; ++it) {
    double Hn[21], gn[6], cn;
    view_linearise(lm.trial, cam, obj, N, observation, complete != 0, mcba::WholeView{}, Hn, gn, cn);
    view_lm_decide(lm, Hn, gn, cn);
    if (lm.done) break;
    view_lm_step(lm);
  }
  bool ok = complete && !lm.failed;
  for (int i = 0; i < 6; ++i) { pose[i] = lm.pose[i]; ok = ok && pnp_finite(lm.pose[i]); }
  *cost = lm.cost;
  if (!complete) return 0;
  return ok ? lm.evals : -1;
}
// Zhang's closed form for one camera: H (n x 9) homographies in pixels (a NaN one is skipped), image size -> K4 (fx fy cx cy); returns 1 if the
// closed form was used, 0 for the fallback
int hc_zhang(int n, const double* H, double w, double h, double* K4) {
  double M[21];
  for (int i = 0; i < 21; ++i) M[i] = 0.0;
  const double s0 = w > h ? w : h;
  int used = 0;
  for (int v = 0; v < n; ++v) {
    bool ok = true;
    for (int i = 0; i < 9; ++i) ok = ok && pnp_finite(H[9 * v + i]);
    if (!ok) continue;
    zhang_accumulate(H + 9 * v, 0.5 * (w - 1.0), 0.5 * (h - 1.0), 1.0 / s0, M);
    ++used;
  }
  return zhang_solve(M, used, w, h, K4) ? 1 : 0;
}
// rotation matrix -> rotation vector as the pose-graph kernels compute it (clamped arccos)
void hc_rotvec(const double* R, double* w) { rotvec_from_matrix(R, w); }
}
