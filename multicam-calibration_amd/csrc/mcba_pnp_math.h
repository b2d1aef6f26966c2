// mcba_pnp_math.h -- the per-view arithmetic of csrc/mcba_pnp.hip (calibrate()'s homography start, pose start and per-view Levenberg-Marquardt:
// what the reference asks of cv2.calibrateCamera's closed form and of cv2.solvePnP, calibration.py:68, :108), written once for the GPU kernel
// (one lane = one view; the kernel owns the wave-uniform loops) and for the host harness (tests/hostcheck/hostcheck.cpp: the same text compiled
// with g++, also under ASan + UBSan, checked against oracle/calibration_oracle.py in the GPU-less tier).
#pragma once
#include "mcba_math.h"

namespace mcba {

MCBA_HD bool pnp_finite(double v) { return fabs(v) < 1.7e308; }   // (false for NaN and +-inf)

struct Cam9 { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };

MCBA_HD Cam9 load_cam9(const double* __restrict__ p) { return Cam9{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }

// pixel -> undistorted normalised coordinates: OpenCV's undistortPoints iteration x <- (x_d - tangential(x)) / radial(x)
MCBA_HD void undistort_norm(double u, double v, const Cam9& k, double ifx, double ify, int iters, double& x, double& y) {
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
MCBA_HD void chol3(const double* A, double eps, double* L) {
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
MCBA_HD void fwd3(const double* L, const double* b, double* z) {   // L z = b
  z[0] = b[0] / L[0];
  z[1] = (b[1] - L[1] * z[0]) / L[3];
  z[2] = (b[2] - L[2] * z[0] - L[4] * z[1]) / L[5];
}
MCBA_HD void bwd3(const double* L, const double* z, double* y) {   // L^T y = z
  y[2] = z[2] / L[5];
  y[1] = (z[1] - L[4] * y[2]) / L[3];
  y[0] = (z[0] - L[1] * y[1] - L[2] * y[2]) / L[0];
}
MCBA_HD double sym3(const double* S, int i, int j) {
  const int a = i < j ? i : j, b = i < j ? j : i;
  return S[a == 0 ? b : (a == 1 ? 2 + b : 5)];
}

// ---- packed symmetric N x N (upper triangle row-major): Cholesky solve in registers
template <int N>
MCBA_HD constexpr int tri(int i, int j) { return i * N - (i * (i - 1)) / 2 + (j - i); }
template <int N>
MCBA_HD bool chol_solve(double* A, double* b) {
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
MCBA_HD void rotvec_from_matrix(const double* R, double* w) {
  const double v0 = R[7] - R[5], v1 = R[2] - R[6], v2 = R[3] - R[1];
  double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  const double th = acos(c);
  double n = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
  n = n == 0.0 ? 1.0 : n;
  w[0] = v0 * th / n; w[1] = v1 * th / n; w[2] = v2 * th / n;
}

// projection of one board point and its derivatives with respect to the pose: Xc = R X + t, Q_k X = d(R X)/dw_k
struct PoseLin {
  double R[9], t[3], Q[27];
};
MCBA_HD void make_pose_lin(const double* pose, PoseLin& pl) {
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


// ---- how a view's points are dealt out.  WholeView: one lane (or the host's plain loop) walks all N points.  The kernel's QuadView
// (csrc/mcba_pnp.hip) gives the points p = part, part + 4, ... to four neighbouring lanes and adds the partial sums up with two DPP steps, after
// which the four lanes hold bit-identical totals and run the serial arithmetic (factorisations, the LM decision) redundantly and in step.
struct WholeView {
  static constexpr int parts = 1;
  MCBA_HD int part() const { return 0; }
  MCBA_HD double sum(double v) const { return v; }
  MCBA_HD bool all(bool b) const { return b; }
};

// ---- Hartley normalisation of a view's image points from ONE pass: sums of the coordinates relative to the first point (so that
// sum d^2 / N - |mean d|^2 cancels the spread against itself, not against the offset of the board in the image).  image_point(p, x, y, present)
// hands out point p (undistorted normalised coordinates, or pixels).  Out: complete (every scalar present), centroid, sqrt(2) / rms distance.
template <class Fetch, class Split>
MCBA_HD void view_normalisation(Fetch& image_point, int N, bool in_range, const Split& sp, bool& complete, double& mx, double& my, double& ss) {
  complete = in_range;
  double x0, y0; bool pr0;
  image_point(0, x0, y0, pr0);
  complete = complete && pr0;
  double sx = 0.0, sy = 0.0, sq = 0.0;
  for (int p = 1 + sp.part(); p < N; p += Split::parts) {
    double x, y; bool pr;
    image_point(p, x, y, pr);
    complete = complete && pr;
    const double dx = x - x0, dy = y - y0;
    sx += dx; sy += dy;
    sq = fma(dx, dx, fma(dy, dy, sq));
  }
  complete = sp.all(complete);
  sx = sp.sum(sx); sy = sp.sum(sy); sq = sp.sum(sq);
  const double inv_n = 1.0 / N, ax = sx * inv_n, ay = sy * inv_n;
  mx = x0 + ax; my = y0 + ay;
  const double ms = sq - N * (ax * ax + ay * ay);   // = sum |p - mean|^2
  ss = complete && ms > 0.0 ? sqrt(2.0) / sqrt(ms / N) : 1.0;
  if (!complete) { mx = 0.0; my = 0.0; }
}

// ---- the DLT: normal matrix of the rows [p 0 -u p], [0 p -v p] (p = (X, Y, 1) normalised; bn = the board's centroid and scale) as four
// 3 x 3 blocks Spp, -Su, -Sv, Sw, and its block Cholesky factor  [[Spp, 0, -Su], [0, Spp, -Sv], [-Su, -Sv, Sw]] + eps I
struct DltFactor { double L11[6], L31[9], L32[9], L33[6]; };
template <class Fetch, class Split>
MCBA_HD void view_dlt_factor(Fetch& image_point, const double* obj, int N, double bmx, double bmy, double bs, bool complete, double mx, double my, double ss, const Split& sp,
                             DltFactor& F) {
  double Spp[6] = {0, 0, 0, 0, 0, 0}, Su[6] = {0, 0, 0, 0, 0, 0}, Sv[6] = {0, 0, 0, 0, 0, 0}, Sw[6] = {0, 0, 0, 0, 0, 0};
  for (int p = sp.part(); p < N; p += Split::parts) {
    double x, y; bool pr;
    image_point(p, x, y, pr);
    const double u = complete ? (x - mx) * ss : 0.0, v = complete ? (y - my) * ss : 0.0;
    const double X = (obj[3 * p] - bmx) * bs, Y = (obj[3 * p + 1] - bmy) * bs;
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
#pragma unroll
  for (int i = 0; i < 6; ++i) { Spp[i] = sp.sum(Spp[i]); Su[i] = sp.sum(Su[i]); Sv[i] = sp.sum(Sv[i]); Sw[i] = sp.sum(Sw[i]); }
  const double eps = 1e-13 * (2.0 * (Spp[0] + Spp[3] + Spp[5]) + Sw[0] + Sw[3] + Sw[5]) / 9.0;
  chol3(Spp, eps, F.L11);
#pragma unroll
  for (int i = 0; i < 3; ++i) {   // row i of L31: L11 (L31 row i)^T = (-Su row i)^T
    const double bu[3] = {-sym3(Su, i, 0), -sym3(Su, i, 1), -sym3(Su, i, 2)};
    const double bv[3] = {-sym3(Sv, i, 0), -sym3(Sv, i, 1), -sym3(Sv, i, 2)};
    fwd3(F.L11, bu, F.L31 + 3 * i);
    fwd3(F.L11, bv, F.L32 + 3 * i);
  }
  double T[6];
  int q = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j, ++q) {
      double s = Sw[q];
#pragma unroll
      for (int k = 0; k < 3; ++k) s -= F.L31[3 * i + k] * F.L31[3 * j + k] + F.L32[3 * i + k] * F.L32[3 * j + k];
      T[q] = s;
    }
  chol3(T, eps, F.L33);
}
// one step of inverse iteration for the smallest eigenvector: h <- normalise(M^-1 h), sign kept; returns max |change|
MCBA_HD double dlt_inverse_iteration(const DltFactor& F, double* h) {
  double z1[3], z2[3], z3[3], r3[3], y[9];
  fwd3(F.L11, h, z1);
  fwd3(F.L11, h + 3, z2);
#pragma unroll
  for (int i = 0; i < 3; ++i) r3[i] = h[6 + i] - (F.L31[3 * i] * z1[0] + F.L31[3 * i + 1] * z1[1] + F.L31[3 * i + 2] * z1[2]) - (F.L32[3 * i] * z2[0] + F.L32[3 * i + 1] * z2[1] + F.L32[3 * i + 2] * z2[2]);
  fwd3(F.L33, r3, z3);
  bwd3(F.L33, z3, y + 6);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    z1[i] -= F.L31[i] * y[6] + F.L31[3 + i] * y[7] + F.L31[6 + i] * y[8];
    z2[i] -= F.L32[i] * y[6] + F.L32[3 + i] * y[7] + F.L32[6 + i] * y[8];
  }
  bwd3(F.L11, z1, y);
  bwd3(F.L11, z2, y + 3);
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
  return diff;
}
MCBA_HD void dlt_start_vector(double* h) {   // (weight on h33, which no admissible homography of centred data lacks)
  const double h0[9] = {0.1, 0.03, 0.02, -0.03, 0.1, 0.01, 0.02, 0.01, 1.0};
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = h0[i];
}
// H = Tu^-1 Hn TX, scaled to H22 = 1 (a division, as numpy's H / H[2, 2]: H22 comes out as exactly 1)
MCBA_HD void homography_denormalise(const double* h, double bmx, double bmy, double bs, double mx, double my, double ss, double* H) {
  double G[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    G[3 * i] = h[3 * i] * bs;
    G[3 * i + 1] = h[3 * i + 1] * bs;
    G[3 * i + 2] = h[3 * i + 2] - bs * (bmx * h[3 * i] + bmy * h[3 * i + 1]);
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
  for (int i = 0; i < 9; ++i) H[i] /= h22;
}
// pose from a homography of normalised coordinates (K = I): H = lam [r1 r2 t]; nearest rotation = the orthogonal polar factor
// (det = |r1 x r2|^2 >= 0) by Newton's iteration X <- (X + X^-T) / 2; rotation vector by the reference's rodrigues_inv
MCBA_HD void pose_from_homography(const double* H, double* pose) {
  const double n0 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n1 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
  double lam = 2.0 / (n0 + n1);
  lam = H[8] < 0.0 ? -lam : lam;
  double X[9];   // columns r1, r2, r1 x r2 (row-major 3 x 3)
  X[0] = H[0] * lam; X[3] = H[3] * lam; X[6] = H[6] * lam;
  X[1] = H[1] * lam; X[4] = H[4] * lam; X[7] = H[7] * lam;
  X[2] = X[3] * X[7] - X[6] * X[4];
  X[5] = X[6] * X[1] - X[0] * X[7];
  X[8] = X[0] * X[4] - X[3] * X[1];
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

// ---- Zhang's closed form for a camera matrix from the board-plane homographies of its views (the start cv2.calibrateCamera computes before it
// refines, calibration.py:68): every view gives two rows v_01 and v_00 - v_11 of a 6-column system V b = 0 in the image of the absolute conic
// b = (B11 B12 B22 B13 B23 B33); one more row holds the skew at zero.  The null vector is the eigenvector of the smallest eigenvalue of V^T V
// (6 x 6, accumulated view by view; cyclic Jacobi, which finds the small eigenvalues of a positive semi-definite matrix to high relative
// accuracy).  Coordinates of order 1: x' = (x - (w - 1) / 2) / max(w, h).
// One view's contribution to the packed upper triangle M (21) of V^T V.  H row-major 3 x 3 in pixels.
MCBA_HD void zhang_accumulate(const double* H, double ox, double oy, double is0, double* M) {
  double a[3], b[3];   // columns 0 and 1 of N H
  a[0] = (H[0] - ox * H[6]) * is0; a[1] = (H[3] - oy * H[6]) * is0; a[2] = H[6];
  b[0] = (H[1] - ox * H[7]) * is0; b[1] = (H[4] - oy * H[7]) * is0; b[2] = H[7];
  const double nrm = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { a[i] *= nrm; b[i] *= nrm; }
  const double r0[6] = {a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[1] * b[1], a[2] * b[0] + a[0] * b[2], a[2] * b[1] + a[1] * b[2], a[2] * b[2]};
  const double r1[6] = {a[0] * a[0] - b[0] * b[0], 2.0 * (a[0] * a[1] - b[0] * b[1]), a[1] * a[1] - b[1] * b[1],
                        2.0 * (a[2] * a[0] - b[2] * b[0]), 2.0 * (a[2] * a[1] - b[2] * b[1]), a[2] * a[2] - b[2] * b[2]};
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) M[tri<6>(i, j)] = fma(r0[i], r0[j], fma(r1[i], r1[j], M[tri<6>(i, j)]));
}
// eigenvector of the smallest eigenvalue of a symmetric 6 x 6 matrix (packed upper triangle, not changed): cyclic Jacobi rotations
MCBA_HD void smallest_eigenvector6(const double* M, double* vec) {
  double A[6][6], V[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) { A[i][j] = M[i <= j ? tri<6>(i, j) : tri<6>(j, i)]; V[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, dia = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dia = fma(A[i][i], A[i][i], dia);
#pragma unroll
      for (int j = i + 1; j < 6; ++j) off = fma(A[i][j], A[i][j], off);
    }
    if (!(off > 1e-34 * dia)) break;
#pragma unroll
    for (int p = 0; p < 5; ++p)
#pragma unroll
      for (int q = p + 1; q < 6; ++q) {
        const double apq = A[p][q];
        const bool rotate = apq != 0.0;
        const double theta = rotate ? (A[q][q] - A[p][p]) / (2.0 * apq) : 0.0;
        const double t = rotate ? (theta < 0.0 ? -1.0 : 1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0))) : 0.0;
        const double c = 1.0 / sqrt(fma(t, t, 1.0)), sn = t * c;
#pragma unroll
        for (int k = 0; k < 6; ++k) {   // A <- A J (columns p, q)
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {   // A <- J^T A (rows p, q);  V <- V J
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq;
          V[k][q] = sn * vkp + c * vkq;
        }
      }
  }
  double best = A[0][0];
#pragma unroll
  for (int k = 0; k < 6; ++k) vec[k] = V[k][0];
#pragma unroll
  for (int j = 1; j < 6; ++j) {
    const bool take = A[j][j] < best;
    best = take ? A[j][j] : best;
#pragma unroll
    for (int k = 0; k < 6; ++k) vec[k] = take ? V[k][j] : vec[k];
  }
}
// K (fx fy cx cy) of an image of w x h pixels from M = V^T V over n usable views (the skew row is added here, weighted by n as the views'
// rows are by their count); the fallback f = max(w, h), c = the image centre when the views do not constrain it (fewer than two, or an
// estimate that is not positive definite).  Returns whether the closed form was used.
MCBA_HD bool zhang_solve(const double* M, int n, double w, double h, double* K4) {
  const double s0 = w > h ? w : h, ox = 0.5 * (w - 1.0), oy = 0.5 * (h - 1.0);
  K4[0] = s0; K4[1] = s0; K4[2] = ox; K4[3] = oy;
  if (n < 2) return false;
  double Ms[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) Ms[i] = M[i];
  Ms[tri<6>(1, 1)] += (double)n * (double)n;
  double b[6];
  smallest_eigenvector6(Ms, b);
  const double b11 = b[0], b12 = b[1], b22 = b[2], b13 = b[3], b23 = b[4], b33 = b[5];
  const double den = b11 * b22 - b12 * b12;
  const double v0 = (b12 * b13 - b11 * b23) / den;
  const double lam = b33 - (b13 * b13 + v0 * (b12 * b13 - b11 * b23)) / b11;
  const double a2 = lam / b11, b2 = lam * b11 / den;
  const bool good = den != 0.0 && b11 != 0.0 && pnp_finite(a2) && pnp_finite(b2) && pnp_finite(v0) && a2 > 0.0 && b2 > 0.0;
  if (!good) return false;
  const double alpha = sqrt(a2), beta = sqrt(b2), u0 = -b13 * alpha * alpha / lam;
  if (!(pnp_finite(u0))) return false;
  K4[0] = s0 * alpha; K4[1] = s0 * beta; K4[2] = fma(s0, u0, ox); K4[3] = fma(s0, v0, oy);
  return true;
}

// ---- one linearisation of a view's pixel reprojection error at `trial` (five-coefficient model): Gauss-Newton block (packed upper triangle),
// gradient J^T e, cost 0.5 sum e^2.  observation(p, u, v) hands out the detection of point p.
template <class Obs, class Split>
MCBA_HD void view_linearise(const double* trial, const Cam9& cam, const double* obj, int N, Obs& observation, bool complete, const Split& sp, double* Hn, double* gn, double& cn) {
  PoseLin pl;
  make_pose_lin(trial, pl);
#pragma unroll
  for (int i = 0; i < 21; ++i) Hn[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) gn[i] = 0.0;
  cn = 0.0;
  for (int p = sp.part(); p < N; p += Split::parts) {
    double ou, ov;
    observation(p, ou, ov);
    const double Xo[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
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
    const double eu = complete ? fma(cam.fx, xd, cam.cx) - ou : 0.0, ev = complete ? fma(cam.fy, yd, cam.cy) - ov : 0.0;
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
#pragma unroll
  for (int i = 0; i < 21; ++i) Hn[i] = sp.sum(Hn[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) gn[i] = sp.sum(gn[i]);
  cn = 0.5 * sp.sum(cn);
}

// ---- the per-view Levenberg-Marquardt state and its decision: after a linearisation at `trial`, accept / reject, Nielsen's damping, scipy's
// tests at tight tolerances, and the next trial point.  Every view its own damping; a view that is done keeps running arithmetic on a zero step.
struct ViewLM {
  double pose[6], trial[6], step[6], Hc[21], gc[6];
  double cost, lam, nu;
  bool done, failed, first;
  int evals;
};
MCBA_HD void view_lm_init(ViewLM& s, const double* pose0, bool complete) {
#pragma unroll
  for (int i = 0; i < 6; ++i) { s.pose[i] = pose0[i]; s.trial[i] = pose0[i]; s.step[i] = 0.0; s.gc[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < 21; ++i) s.Hc[i] = 0.0;
  s.cost = 0.0; s.lam = 1e-3; s.nu = 2.0;
  s.done = !complete; s.failed = !complete; s.first = true;
  s.evals = 0;
}
MCBA_HD void view_lm_decide(ViewLM& s, const double* Hn, const double* gn, double cn) {
  s.evals += s.done ? 0 : 1;
  const bool finite_new = pnp_finite(cn);
  bool accept;
  double gain = 0.0, ratio = 1.0;
  if (s.first) {
    accept = !s.done;
    s.failed = s.failed || !finite_new;
    s.done = s.done || !finite_new;
  } else {
    double pred = 0.0;   // the model's reduction: -g.d - d H d / 2
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double hd = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) hd = fma(s.Hc[i <= j ? tri<6>(i, j) : tri<6>(j, i)], s.step[j], hd);
      pred -= s.step[i] * fma(0.5, hd, s.gc[i]);
    }
    gain = s.cost - cn;
    accept = !s.done && finite_new && gain >= -1e-13 * s.cost;   // (a loss inside the cost's own rounding level is a converged lane, not an uphill step)
    ratio = pred > 0.0 ? gain / pred : (gain > 0.0 ? 1.0 : 0.0);
  }
  if (accept) {
    double sn = 0.0, xn = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { sn = fma(s.step[i], s.step[i], sn); xn = fma(s.trial[i], s.trial[i], xn); s.pose[i] = s.trial[i]; s.gc[i] = gn[i]; }
#pragma unroll
    for (int i = 0; i < 21; ++i) s.Hc[i] = Hn[i];
    if (!s.first) {
      const double t = 2.0 * ratio - 1.0;
      s.lam = fmax(s.lam * fmax(1.0 / 3.0, 1.0 - t * t * t), 1e-12);
      s.nu = 2.0;
      // scipy's tests (common.py:705-717) at tight tolerances: the relative gain (at 1e-13 the gain ratio is rounding noise: not asked for), the step
      const bool f_small = fabs(gain) <= 1e-13 * s.cost;
      const bool x_small = sqrt(sn) <= 1e-12 * (1e-12 + sqrt(xn));
      s.done = s.done || f_small || x_small;
    }
    s.cost = cn;
    double gmax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) gmax = fmax(gmax, fabs(s.gc[i]));
    s.done = s.done || gmax <= 1e-10 || s.cost == 0.0;
  } else if (!s.done) {
    s.lam *= s.nu; s.nu *= 2.0;
    s.done = s.done || s.lam > 1e12;
  }
  s.first = false;
}
// next trial point: (H + lam diag H) d = -g
MCBA_HD void view_lm_step(ViewLM& s) {
  double A[21], b[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) A[i] = s.Hc[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double dgn = s.Hc[tri<6>(i, i)];
    A[tri<6>(i, i)] = s.done ? 1.0 : fma(s.lam, dgn > 0.0 ? dgn : 1.0, dgn);
    b[i] = s.done ? 0.0 : -s.gc[i];
  }
  const bool ok = chol_solve<6>(A, b);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    s.step[i] = ok ? b[i] : 0.0;
    s.trial[i] = s.pose[i] + s.step[i];
  }
  if (!ok && !s.done) { s.lam *= s.nu; s.nu *= 2.0; s.done = s.lam > 1e12; }   // (a zero step is then "accepted" with no gain; the damping has grown)
}

}  // namespace mcba
