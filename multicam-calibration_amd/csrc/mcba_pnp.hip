// mcba_pnp.hip -- calibrate()'s per-view work and its pose graph on the GPU (reference: multicam_calibration/calibration.py).
//
//   k_view_complete   per (camera, frame): is every one of the 2 N scalars of the detection present (calibration.py:55, :107)
//   k_pnp             lane = one view (camera, frame): what the reference asks of OpenCV per view --
//                       MODE_HOMOGRAPHY  the board-plane -> image homography (the closed-form start of cv2.calibrateCamera, :68): Hartley-
//                                        normalised DLT, the null vector of the 2N x 9 system found as the smallest eigenvector of its
//                                        9 x 9 normal matrix by inverse iteration on a block Cholesky factor (three 3 x 3 blocks: the rows
//                                        of h1 and h2 do not couple);
//                       MODE_POSE        cv2.solvePnP (:108) for a planar board: undistort (OpenCV's fixed point), homography on the
//                                        normalised coordinates, pose from the homography (polar factor by Newton's iteration), then
//                                        Levenberg-Marquardt on the reprojection error in pixels with the full five-coefficient model,
//                                        every lane its own damping and its own stopping test.
//                     Dense form: blockIdx.y = camera, lane = frame, observations read from obs_t [C][N][Fpad] (one coalesced 1 KiB load per
//                     point and wavefront); list form: an explicit list of (camera, frame) views (the <= 100 sampled views per camera of
//                     get_intrinsics).
//   k_pose_pairs      calibration.py:116-143: T2 T1^-1 per common frame of a camera pair, as 6-vectors (their medians: the radix select of
//                     mcba_diag.hip on order-preserving keys);  k_pose_consensus: calibration.py:239-277, the median over cameras in the lane.
// FP64 throughout.  Nothing here is in the LM loop of bundle_adjust(); it is the initialiser that produces its inputs.
#include <hip/hip_runtime.h>

#include "mcba_kernels.h"
#include "mcba_math.h"

namespace mcba {

namespace {

struct Cam9 { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };

__device__ __forceinline__ Cam9 load_cam9(const double* __restrict__ p) { return Cam9{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }

// pixel -> undistorted normalised coordinates: OpenCV's undistortPoints iteration x <- (x_d - tangential(x)) / radial(x)
__device__ __forceinline__ void undistort_norm(double u, double v, const Cam9& k, double ifx, double ify, int iters, double& x, double& y) {
  // (reciprocals by fast_rcp -- hardware estimate + one cubic step, 1.1e-16: a quarter of the instructions of the two IEEE divisions per round,
  //  which were a fifth of this kernel's instruction stream; ifx / ify are the caller's 1 / fx, 1 / fy)
  const double x0 = (u - k.cx) * ifx, y0 = (v - k.cy) * ify;
  x = x0; y = y0;
  for (int it = 0; it < iters; ++it) {
    const double s = x * x + y * y;
    const double id = fast_rcp(1.0 + s * (k.k1 + s * (k.k2 + s * k.k3)));
    const double dx = 2.0 * k.p1 * x * y + k.p2 * (s + 2.0 * x * x);
    const double dy = k.p1 * (s + 2.0 * y * y) + 2.0 * k.p2 * x * y;
    x = (x0 - dx) * id;
    y = (y0 - dy) * id;
  }
}

// ---- 3 x 3 symmetric matrices packed as (00 01 02 11 12 22); lower Cholesky factors in the same slots (L00 L10 L20 L11 L21 L22)
__device__ __forceinline__ void chol3(const double* A, double eps, double* L) {
  const double floor_ = eps > 0.0 ? eps : 1e-300;
  double d = A[0] + eps;
  L[0] = sqrt(d > floor_ ? d : floor_);
  L[1] = A[1] / L[0];
  L[2] = A[2] / L[0];
  d = A[3] + eps - L[1] * L[1];
  L[3] = sqrt(d > floor_ ? d : floor_);
  L[4] = (A[4] - L[2] * L[1]) / L[3];
  d = A[5] + eps - L[2] * L[2] - L[4] * L[4];
  L[5] = sqrt(d > floor_ ? d : floor_);
}
__device__ __forceinline__ void fwd3(const double* L, const double* b, double* z) {   // L z = b
  z[0] = b[0] / L[0];
  z[1] = (b[1] - L[1] * z[0]) / L[3];
  z[2] = (b[2] - L[2] * z[0] - L[4] * z[1]) / L[5];
}
__device__ __forceinline__ void bwd3(const double* L, const double* z, double* y) {   // L^T y = z
  y[2] = z[2] / L[5];
  y[1] = (z[1] - L[4] * y[2]) / L[3];
  y[0] = (z[0] - L[1] * y[1] - L[2] * y[2]) / L[0];
}
__device__ __forceinline__ double sym3(const double* S, int i, int j) {
  const int a = i < j ? i : j, b = i < j ? j : i;
  return S[a == 0 ? b : (a == 1 ? 2 + b : 5)];
}

// ---- packed symmetric N x N (upper triangle row-major): Cholesky solve in registers
template <int N>
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }
template <int N>
__device__ __forceinline__ bool chol_solve(double* A, double* b) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = i; j < N; ++j) {
      double s = A[tri<N>(i, j)];
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(-A[tri<N>(k, i)], A[tri<N>(k, j)], s);
      if (j == i) {
        ok = ok && s > 0.0;
        A[tri<N>(i, i)] = sqrt(s > 0.0 ? s : 1.0);
      } else {
        A[tri<N>(i, j)] = s / A[tri<N>(i, i)];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = fma(-A[tri<N>(k, i)], b[k], s);
    b[i] = s / A[tri<N>(i, i)];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s = fma(-A[tri<N>(i, k)], b[k], s);
    b[i] = s / A[tri<N>(i, i)];
  }
  return ok;
}

// rotation matrix -> rotation vector by the reference's formula (geometry.py:38-56: theta = arccos((tr - 1) / 2), axis from the skew part);
// the arccos argument is clamped (the reference returns NaN when rounding pushes it past 1)
__device__ __forceinline__ void rotvec_from_matrix(const double* R, double* w) {
  const double v0 = R[7] - R[5], v1 = R[2] - R[6], v2 = R[3] - R[1];
  double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  const double th = acos(c);
  double n = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
  n = n == 0.0 ? 1.0 : n;
  w[0] = v0 * th / n; w[1] = v1 * th / n; w[2] = v2 * th / n;
}

constexpr int MODE_HOMOGRAPHY = 0, MODE_POSE = 1;

struct PnpArgs {
  const double2* obs_t;     // [C][N][Fpad]
  const double* obj;        // (N, 3)
  const double* intr9;      // [C][9] fx fy cx cy k1 k2 p1 p2 k3 (MODE_POSE)
  const int* views;         // list form: nviews x (camera, frame); nullptr = dense
  double bmx, bmy, bs;      // Hartley normalisation of the board's XY: centroid, sqrt(2) / rms distance
  int nviews, C, F, N, Fpad;
  int und_iters, lm_iters;
  double* out;              // list form: nviews x 9 (homography) / nviews x 6 (pose); dense: (C, F, 6) or nullptr
  double* poses_t;          // dense MODE_POSE: [C][6][Fpad], NaN where no pose (device-resident input of the pose graph) or nullptr
  unsigned char* valid;     // per view (list: [nviews], dense: [C][F]): 1 = a pose / homography came out, or nullptr
  unsigned char* nit;       // LM evaluations the view took (diagnostics), same indexing, or nullptr
};

// projection of one board point and its derivatives with respect to the pose: Xc = R X + t, Q_k X = d(R X)/dw_k
struct PoseLin {
  double R[9], t[3], Q[27];
};
__device__ __forceinline__ void make_pose_lin(const double* pose, PoseLin& pl) {
  double Jr[9];
  rot_and_jr(pose, pl.R, Jr);
  pl.t[0] = pose[3]; pl.t[1] = pose[4]; pl.t[2] = pose[5];
  // d(R(w) X)/dw_k = R (Jr e_k x X) = R [j_k]x X with j_k = column k of the right Jacobian
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double j0 = Jr[k], j1 = Jr[3 + k], j2 = Jr[6 + k];
    const double S[9] = {0.0, -j2, j1, j2, 0.0, -j0, -j1, j0, 0.0};
    mm33(pl.R, S, pl.Q + 9 * k);
  }
}

template <int MODE, bool DENSE>
__global__ __launch_bounds__(64) void k_pnp(PnpArgs a) {
  const int lane = threadIdx.x;
  int c, f, vi;
  bool in_range;
  if (DENSE) {
    c = blockIdx.y; f = blockIdx.x * 64 + lane; vi = 0;
    in_range = f < a.F;
  } else {
    vi = blockIdx.x * 64 + lane;
    in_range = vi < a.nviews;
    c = in_range ? a.views[2 * vi] : 0;
    f = in_range ? a.views[2 * vi + 1] : 0;
  }
  const int N = a.N;
  const size_t Fpad = (size_t)a.Fpad;
  const double2* op = a.obs_t + (size_t)c * N * Fpad + (DENSE ? (size_t)f : (size_t)(in_range ? f : 0));
  Cam9 cam{1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (MODE == MODE_POSE) cam = load_cam9(a.intr9 + 9 * c);
  const double ifx = 1.0 / cam.fx, ify = 1.0 / cam.fy;
  auto image_point = [&](int p, double& x, double& y, bool& present) {
    const double2 o = op[(size_t)p * Fpad];
    present = o.x == o.x && o.y == o.y;
    if (MODE == MODE_POSE) undistort_norm(o.x, o.y, cam, ifx, ify, a.und_iters, x, y);
    else { x = o.x; y = o.y; }
  };

  // ---- pass 1: complete?  centroid and rms distance of the image points (Hartley) from ONE pass: sums of the coordinates relative to the
  // first point (so that sum d^2 / N - |mean d|^2 cancels the spread against itself, not against the offset of the board in the image)
  bool complete = in_range;
  double mx, my, ms;
  {
    double x0, y0; bool pr0;
    image_point(0, x0, y0, pr0);
    complete = complete && pr0;
    double sx = 0.0, sy = 0.0, sq = 0.0;
    for (int p = 1; p < N; ++p) {
      double x, y; bool pr;
      image_point(p, x, y, pr);
      complete = complete && pr;
      const double dx = x - x0, dy = y - y0;
      sx += dx; sy += dy;
      sq = fma(dx, dx, fma(dy, dy, sq));
    }
    const double inv_n = 1.0 / N, ax = sx * inv_n, ay = sy * inv_n;
    mx = x0 + ax; my = y0 + ay;
    ms = sq - N * (ax * ax + ay * ay);   // = sum |p - mean|^2
  }
  const double ss = complete && ms > 0.0 ? sqrt(2.0) / sqrt(ms / N) : 1.0;
  if (!complete) { mx = 0.0; my = 0.0; }

  // ---- pass 3: the normal matrix of the DLT rows [p 0 -u p], [0 p -v p] (p = (X, Y, 1) normalised): blocks Spp, -Su, -Sv, Sw
  double Spp[6] = {0, 0, 0, 0, 0, 0}, Su[6] = {0, 0, 0, 0, 0, 0}, Sv[6] = {0, 0, 0, 0, 0, 0}, Sw[6] = {0, 0, 0, 0, 0, 0};
  for (int p = 0; p < N; ++p) {
    double x, y; bool pr;
    image_point(p, x, y, pr);
    const double u = complete ? (x - mx) * ss : 0.0, v = complete ? (y - my) * ss : 0.0;
    const double X = (a.obj[3 * p] - a.bmx) * a.bs, Y = (a.obj[3 * p + 1] - a.bmy) * a.bs;
    const double pp[6] = {X * X, X * Y, X, Y * Y, Y, 1.0};
    const double w = u * u + v * v;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      Spp[i] += pp[i];
      Su[i] = fma(u, pp[i], Su[i]);
      Sv[i] = fma(v, pp[i], Sv[i]);
      Sw[i] = fma(w, pp[i], Sw[i]);
    }
  }
  // block Cholesky of M + eps I:  [[Spp, 0, -Su], [0, Spp, -Sv], [-Su, -Sv, Sw]]
  const double eps = 1e-13 * (2.0 * (Spp[0] + Spp[3] + Spp[5]) + Sw[0] + Sw[3] + Sw[5]) / 9.0;
  double L11[6], L31[9], L32[9], L33[6];
  chol3(Spp, eps, L11);
#pragma unroll
  for (int i = 0; i < 3; ++i) {   // row i of L31: L11 (L31 row i)^T = (-Su row i)^T
    const double bu[3] = {-sym3(Su, i, 0), -sym3(Su, i, 1), -sym3(Su, i, 2)};
    const double bv[3] = {-sym3(Sv, i, 0), -sym3(Sv, i, 1), -sym3(Sv, i, 2)};
    fwd3(L11, bu, L31 + 3 * i);
    fwd3(L11, bv, L32 + 3 * i);
  }
  {
    double T[6];
    int q = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j, ++q) {
        double s = Sw[q];
#pragma unroll
        for (int k = 0; k < 3; ++k) s -= L31[3 * i + k] * L31[3 * j + k] + L32[3 * i + k] * L32[3 * j + k];
        T[q] = s;
      }
    chol3(T, eps, L33);
  }
  // inverse iteration for the smallest eigenvector (the start has weight on h33, which no admissible homography of centred data lacks)
  double h[9] = {0.1, 0.03, 0.02, -0.03, 0.1, 0.01, 0.02, 0.01, 1.0};
  for (int it = 0; it < 60; ++it) {
    double z1[3], z2[3], z3[3], r3[3], y[9];
    fwd3(L11, h, z1);
    fwd3(L11, h + 3, z2);
#pragma unroll
    for (int i = 0; i < 3; ++i) r3[i] = h[6 + i] - (L31[3 * i] * z1[0] + L31[3 * i + 1] * z1[1] + L31[3 * i + 2] * z1[2]) - (L32[3 * i] * z2[0] + L32[3 * i + 1] * z2[1] + L32[3 * i + 2] * z2[2]);
    fwd3(L33, r3, z3);
    bwd3(L33, z3, y + 6);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      z1[i] -= L31[i] * y[6] + L31[3 + i] * y[7] + L31[6 + i] * y[8];
      z2[i] -= L32[i] * y[6] + L32[3 + i] * y[7] + L32[6 + i] * y[8];
    }
    bwd3(L11, z1, y);
    bwd3(L11, z2, y + 3);
    double nn = 0.0, dot = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { nn = fma(y[i], y[i], nn); dot = fma(y[i], h[i], dot); }
    const double sc = (dot < 0.0 ? -1.0 : 1.0) / sqrt(nn > 0.0 ? nn : 1.0);
    double diff = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double hn = y[i] * sc;
      diff = fmax(diff, fabs(hn - h[i]));
      h[i] = hn;
    }
    const bool more = complete && !(diff <= 4e-16);
    if (!__any(more)) break;
  }
  // H = Tu^-1 Hn TX, scaled to H22 = 1
  double H[9];
  {
    double G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      G[3 * i] = h[3 * i] * a.bs;
      G[3 * i + 1] = h[3 * i + 1] * a.bs;
      G[3 * i + 2] = h[3 * i + 2] - a.bs * (a.bmx * h[3 * i] + a.bmy * h[3 * i + 1]);
    }
    const double is = 1.0 / ss;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      H[j] = G[j] * is + mx * G[6 + j];
      H[3 + j] = G[3 + j] * is + my * G[6 + j];
      H[6 + j] = G[6 + j];
    }
    const double h22 = H[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] /= h22;   // (a division, as numpy's H / H[2, 2]: H22 comes out as exactly 1)
  }
  const double nan = __builtin_nan("");
  if (MODE == MODE_HOMOGRAPHY) {
    bool ok = complete;
#pragma unroll
    for (int i = 0; i < 9; ++i) ok = ok && isfinite(H[i]);
    if (in_range) {
#pragma unroll
      for (int i = 0; i < 9; ++i) a.out[(size_t)vi * 9 + i] = ok ? H[i] : nan;
      if (a.valid) a.valid[vi] = ok ? 1 : 0;
    }
    return;
  }

  // ---- pose from the homography (coordinates are normalised: K = I): H = lam [r1 r2 t]
  double pose[6];
  {
    const double n0 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n1 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
    double lam = 2.0 / (n0 + n1);
    lam = H[8] < 0.0 ? -lam : lam;
    double X[9];   // columns r1, r2, r1 x r2 (row-major 3 x 3)
    X[0] = H[0] * lam; X[3] = H[3] * lam; X[6] = H[6] * lam;
    X[1] = H[1] * lam; X[4] = H[4] * lam; X[7] = H[7] * lam;
    X[2] = X[3] * X[7] - X[6] * X[4];
    X[5] = X[6] * X[1] - X[0] * X[7];
    X[8] = X[0] * X[4] - X[3] * X[1];
    // nearest rotation = the orthogonal polar factor (det X = |r1 x r2|^2 >= 0): Newton's iteration X <- (X + X^-T) / 2
    for (int it = 0; it < 12; ++it) {
      double Cf[9];   // cofactor matrix: X^-T = Cf / det
      Cf[0] = X[4] * X[8] - X[5] * X[7]; Cf[1] = X[5] * X[6] - X[3] * X[8]; Cf[2] = X[3] * X[7] - X[4] * X[6];
      Cf[3] = X[2] * X[7] - X[1] * X[8]; Cf[4] = X[0] * X[8] - X[2] * X[6]; Cf[5] = X[1] * X[6] - X[0] * X[7];
      Cf[6] = X[1] * X[5] - X[2] * X[4]; Cf[7] = X[2] * X[3] - X[0] * X[5]; Cf[8] = X[0] * X[4] - X[1] * X[3];
      const double det = X[0] * Cf[0] + X[1] * Cf[1] + X[2] * Cf[2];
      const double id = 0.5 / det;
#pragma unroll
      for (int i = 0; i < 9; ++i) X[i] = fma(Cf[i], id, 0.5 * X[i]);
    }
    rotvec_from_matrix(X, pose);
    pose[3] = H[2] * lam; pose[4] = H[5] * lam; pose[5] = H[8] * lam;
  }

  // ---- Levenberg-Marquardt on the reprojection error in pixels (five-coefficient model), one problem per lane
  cam = load_cam9(a.intr9 + 9 * c);
  double Hc[21], gc[6], cost = 0.0, lam = 1e-3, nu = 2.0;
  double trial[6], step[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { trial[i] = pose[i]; step[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < 21; ++i) Hc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) gc[i] = 0.0;
  bool done = !complete, failed = !complete, first = true;
  int evals = 0;
  for (int it = 0; it < a.lm_iters; ++it) {
    // linearise at the trial point
    PoseLin pl;
    make_pose_lin(trial, pl);
    double Hn[21], gn[6], cn = 0.0;
#pragma unroll
    for (int i = 0; i < 21; ++i) Hn[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) gn[i] = 0.0;
    for (int p = 0; p < N; ++p) {
      const double2 o = op[(size_t)p * Fpad];
      const double Xo[3] = {a.obj[3 * p], a.obj[3 * p + 1], a.obj[3 * p + 2]};
      double Xc[3];
      mv3(pl.R, Xo, Xc);
      Xc[0] += pl.t[0]; Xc[1] += pl.t[1]; Xc[2] += pl.t[2];
      const double iz = fast_rcp(Xc[2]);
      const double x = Xc[0] * iz, y = Xc[1] * iz;
      const double r2 = x * x + y * y;
      const double rad = 1.0 + r2 * (cam.k1 + r2 * (cam.k2 + r2 * cam.k3));
      const double drad = cam.k1 + r2 * (2.0 * cam.k2 + 3.0 * cam.k3 * r2);
      const double xd = x * rad + 2.0 * cam.p1 * x * y + cam.p2 * (r2 + 2.0 * x * x);
      const double yd = y * rad + cam.p1 * (r2 + 2.0 * y * y) + 2.0 * cam.p2 * x * y;
      const double eu = complete ? fma(cam.fx, xd, cam.cx) - o.x : 0.0, ev = complete ? fma(cam.fy, yd, cam.cy) - o.y : 0.0;
      // d(xd, yd)/d(x, y)
      const double axx = rad + 2.0 * x * x * drad + 2.0 * cam.p1 * y + 6.0 * cam.p2 * x;
      const double axy = 2.0 * x * y * drad + 2.0 * cam.p1 * x + 2.0 * cam.p2 * y;
      const double ayy = rad + 2.0 * y * y * drad + 6.0 * cam.p1 * y + 2.0 * cam.p2 * x;
      // P = d(u, v)/dXc
      const double P0[3] = {cam.fx * axx * iz, cam.fx * axy * iz, -cam.fx * (axx * x + axy * y) * iz};
      const double P1[3] = {cam.fy * axy * iz, cam.fy * ayy * iz, -cam.fy * (axy * x + ayy * y) * iz};
      double ju[6], jv[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double d[3];
        mv3(pl.Q + 9 * k, Xo, d);
        ju[k] = P0[0] * d[0] + P0[1] * d[1] + P0[2] * d[2];
        jv[k] = P1[0] * d[0] + P1[1] * d[1] + P1[2] * d[2];
        ju[3 + k] = P0[k];
        jv[3 + k] = P1[k];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) Hn[tri<6>(i, j)] = fma(ju[i], ju[j], fma(jv[i], jv[j], Hn[tri<6>(i, j)]));
        gn[i] = fma(ju[i], eu, fma(jv[i], ev, gn[i]));
      }
      cn = fma(eu, eu, fma(ev, ev, cn));
    }
    cn *= 0.5;
    evals += done ? 0 : 1;
    // decide
    const bool finite_new = isfinite(cn);
    bool accept;
    double gain = 0.0, ratio = 1.0;
    if (first) {
      accept = !done;
      failed = failed || !finite_new;
      done = done || !finite_new;
    } else {
      double pred = 0.0;   // the model's reduction: -g.d - d H d / 2
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double hd = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) hd = fma(Hc[i <= j ? tri<6>(i, j) : tri<6>(j, i)], step[j], hd);
        pred -= step[i] * fma(0.5, hd, gc[i]);
      }
      gain = cost - cn;
      accept = !done && finite_new && gain >= -1e-13 * cost;   // (a loss inside the cost's own rounding level is a converged lane, not an uphill step)
      ratio = pred > 0.0 ? gain / pred : (gain > 0.0 ? 1.0 : 0.0);
    }
    if (accept) {
      double sn = 0.0, xn = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) { sn = fma(step[i], step[i], sn); xn = fma(trial[i], trial[i], xn); pose[i] = trial[i]; gc[i] = gn[i]; }
#pragma unroll
      for (int i = 0; i < 21; ++i) Hc[i] = Hn[i];
      if (!first) {
        const double t = 2.0 * ratio - 1.0;
        lam = fmax(lam * fmax(1.0 / 3.0, 1.0 - t * t * t), 1e-12);
        nu = 2.0;
        // scipy's tests (common.py:705-717) at tight tolerances: the relative gain (at 1e-13 the gain ratio is rounding noise: not asked for), the step
        const bool f_small = fabs(gain) <= 1e-13 * cost;
        const bool x_small = sqrt(sn) <= 1e-12 * (1e-12 + sqrt(xn));
        done = done || f_small || x_small;
      }
      cost = cn;
      double gmax = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) gmax = fmax(gmax, fabs(gc[i]));
      done = done || gmax <= 1e-10 || cost == 0.0;
    } else if (!done) {
      lam *= nu; nu *= 2.0;
      done = done || lam > 1e12;
    }
    first = false;
    if (!__any(!done)) break;
    // next trial point: (H + lam diag H) d = -g
    {
      double A[21], b[6];
#pragma unroll
      for (int i = 0; i < 21; ++i) A[i] = Hc[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double dgn = Hc[tri<6>(i, i)];
        A[tri<6>(i, i)] = done ? 1.0 : fma(lam, dgn > 0.0 ? dgn : 1.0, dgn);
        b[i] = done ? 0.0 : -gc[i];
      }
      const bool ok = chol_solve<6>(A, b);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        step[i] = ok ? b[i] : 0.0;
        trial[i] = pose[i] + step[i];
      }
      if (!ok && !done) { lam *= nu; nu *= 2.0; done = lam > 1e12; }   // (a zero step is then "accepted" with no gain; the damping has grown)
    }
  }
  bool ok = complete && !failed;
#pragma unroll
  for (int i = 0; i < 6; ++i) ok = ok && isfinite(pose[i]);
  if (!in_range) {
    if (DENSE && a.poses_t && f < a.Fpad) {
#pragma unroll
      for (int i = 0; i < 6; ++i) a.poses_t[((size_t)c * 6 + i) * Fpad + f] = nan;
    }
    return;
  }
  if (DENSE) {
    const size_t cf = (size_t)c * a.F + f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double v = ok ? pose[i] : nan;
      if (a.out) a.out[cf * 6 + i] = v;
      if (a.poses_t) a.poses_t[((size_t)c * 6 + i) * Fpad + f] = v;
    }
    if (a.valid) a.valid[cf] = ok ? 1 : 0;
    if (a.nit) a.nit[cf] = (unsigned char)(evals > 255 ? 255 : evals);
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) a.out[(size_t)vi * 6 + i] = ok ? pose[i] : nan;
    if (a.valid) a.valid[vi] = ok ? 1 : 0;
    if (a.nit) a.nit[vi] = (unsigned char)(evals > 255 ? 255 : evals);
  }
}

// per (camera, frame): every scalar of the detection present?
__global__ __launch_bounds__(256) void k_view_complete(const double2* __restrict__ obs_t, unsigned char* __restrict__ out, int C, int F, int N, int Fpad) {
  const int c = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const double2* op = obs_t + (size_t)c * N * Fpad + f;
  bool complete = true;
  for (int p = 0; p < N; ++p) {
    const double2 o = op[(size_t)p * Fpad];
    complete = complete && o.x == o.x && o.y == o.y;
  }
  out[(size_t)c * F + f] = complete ? 1 : 0;
}

// pose element (camera c, frame f, coordinate k) of an array with strides (sc, sf, sk): the device-resident [C][6][Fpad] layout of k_pnp
// or a caller's (C, F, 6) array
struct PoseView {
  const double* p;
  size_t sc, sf, sk;
  __device__ __forceinline__ bool load(int c, int f, double* out) const {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      out[k] = p[(size_t)c * sc + (size_t)f * sf + (size_t)k * sk];
      ok = ok && out[k] == out[k];
    }
    return ok;
  }
};

// calibration.py:116-143: per tree edge (c1, c2) and frame both cameras have a pose for: the 6-vector of T2 T1^-1 -> rel[e][k][Fpad]
// (NaN where there is no value: the radix select of the medians skips those)
__global__ __launch_bounds__(64) void k_pose_pairs(PoseView pv, const int* __restrict__ edges, int F, int Fpad, double* __restrict__ rel) {
  const int e = blockIdx.y, f = blockIdx.x * 64 + threadIdx.x;
  const int c1 = edges[2 * e], c2 = edges[2 * e + 1];
  double p1[6], p2[6];
  bool ok = f < F;
  const int fl = ok ? f : 0;
  ok = pv.load(c1, fl, p1) && ok;
  ok = pv.load(c2, fl, p2) && ok;
  double R1[9], R2[9], R[9], w[6];
  rot_only(p1, R1);
  rot_only(p2, R2);
  mmt33(R2, R1, R);          // R2 R1^T
  rotvec_from_matrix(R, w);   // (clamped arccos: a pair of coincident cameras gives the zero rotation where the reference's rodrigues_inv gives NaN)
  double Rt1[3];
  mv3(R, p1 + 3, Rt1);
  w[3] = p2[3] - Rt1[0]; w[4] = p2[4] - Rt1[1]; w[5] = p2[5] - Rt1[2];
  const double nan = __builtin_nan("");
#pragma unroll
  for (int k = 0; k < 6; ++k) rel[((size_t)e * 6 + k) * Fpad + f] = ok ? w[k] : nan;
}

// calibration.py:239-277: every camera's board pose mapped to world coordinates (T_ext^-1 T_pose), nan-median over the cameras per coordinate.
// world: scratch [C][6][Fpad].  Lane = frame; the median by rank counting among the <= C values of the lane (C <= 64).
__global__ __launch_bounds__(64) void k_pose_consensus(PoseView pv, const double* __restrict__ ext, int C, int F, int Fpad, double* __restrict__ world, double* __restrict__ out) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  const bool in_range = f < F;
  const int fl = in_range ? f : 0;
  const double nan = __builtin_nan("");
  for (int c = 0; c < C; ++c) {
    double ps[6], Re[9], Rp[9], R[9], w[6], d[3];
    const bool ok = pv.load(c, fl, ps) && in_range;
    const double* e = ext + 6 * c;
    const double e6[6] = {e[0], e[1], e[2], e[3], e[4], e[5]};
    rot_only(e6, Re);
    rot_only(ps, Rp);
    mtm33(Re, Rp, R);   // Re^T Rp
    rotvec_from_matrix(R, w);
    d[0] = ps[3] - e6[3]; d[1] = ps[4] - e6[4]; d[2] = ps[5] - e6[5];
    mtv3(Re, d, w + 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) world[((size_t)c * 6 + k) * Fpad + f] = ok ? w[k] : nan;
  }
  // (each lane reads back what it wrote itself: no synchronisation needed)
  for (int k = 0; k < 6; ++k) {
    int n = 0;
    for (int c = 0; c < C; ++c) { const double v = world[((size_t)c * 6 + k) * Fpad + f]; n += v == v ? 1 : 0; }
    double lo = nan, hi = nan;
    const int rlo = (n - 1) / 2, rhi = n / 2;
    for (int i = 0; i < C && n > 0; ++i) {
      const double vi = world[((size_t)i * 6 + k) * Fpad + f];
      if (!(vi == vi)) continue;
      int rank = 0;
      for (int j = 0; j < C; ++j) {
        const double vj = world[((size_t)j * 6 + k) * Fpad + f];
        rank += (vj < vi || (vj == vi && j < i)) ? 1 : 0;
      }
      if (rank == rlo) lo = vi;
      if (rank == rhi) hi = vi;
    }
    if (in_range) out[(size_t)f * 6 + k] = n > 0 ? 0.5 * (lo + hi) : nan;
  }
}

}  // namespace

// ---------------------------------------------------------------- launch wrappers
void launch_view_complete(hipStream_t st, const double* obs_t, unsigned char* out, int C, int F, int N, int Fpad) {
  k_view_complete<<<dim3((F + 255) / 256, C), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(obs_t), out, C, F, N, Fpad);
}

void launch_pnp(hipStream_t st, int mode, const double* obs_t, const double* obj, const double* intr9, const int* views, int nviews, const double* bn3, int C, int F, int N, int Fpad, int und_iters, int lm_iters,
                double* out, double* poses_t, unsigned char* valid, unsigned char* nit) {
  PnpArgs a;
  a.obs_t = reinterpret_cast<const double2*>(obs_t); a.obj = obj; a.intr9 = intr9; a.views = views;
  a.bmx = bn3[0]; a.bmy = bn3[1]; a.bs = bn3[2];
  a.nviews = nviews; a.C = C; a.F = F; a.N = N; a.Fpad = Fpad; a.und_iters = und_iters; a.lm_iters = lm_iters;
  a.out = out; a.poses_t = poses_t; a.valid = valid; a.nit = nit;
  if (views) {
    const dim3 grid((nviews + 63) / 64);
    if (mode == MODE_HOMOGRAPHY) k_pnp<MODE_HOMOGRAPHY, false><<<grid, dim3(64), 0, st>>>(a);
    else k_pnp<MODE_POSE, false><<<grid, dim3(64), 0, st>>>(a);
  } else {
    const dim3 grid(Fpad / 64, C);
    k_pnp<MODE_POSE, true><<<grid, dim3(64), 0, st>>>(a);
  }
}

void launch_pose_pairs(hipStream_t st, const double* poses, size_t sc, size_t sf, size_t sk, const int* edges, int n_edges, int F, int Fpad, double* rel) {
  k_pose_pairs<<<dim3(Fpad / 64, n_edges), dim3(64), 0, st>>>(PoseView{poses, sc, sf, sk}, edges, F, Fpad, rel);
}

void launch_pose_consensus(hipStream_t st, const double* poses, size_t sc, size_t sf, size_t sk, const double* ext, int C, int F, int Fpad, double* world, double* out) {
  k_pose_consensus<<<dim3(Fpad / 64), dim3(64), 0, st>>>(PoseView{poses, sc, sf, sk}, ext, C, F, Fpad, world, out);
}

}  // namespace mcba
