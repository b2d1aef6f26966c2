// mcba_calib.hip -- single-camera calibration with OpenCV's FIVE-coefficient distortion model (k1 k2 p1 p2 k3): the per-view normal
// equations that `get_intrinsics(fix_k3=False | zero_tangent_dist=False)` and `estimate_pose` with such coefficients are refined with
// (reference: multicam_calibration/calibration.py:11-71 -> cv2.calibrateCamera, :74-113 -> cv2.solvePnP; this image has no cv2).
//
// The model is OpenCV's published one (calib3d: "Camera Calibration and 3D Reconstruction", distortion section):
//   X_c = R(w) X + t,  x = X_c.x / X_c.z,  y = X_c.y / X_c.z,  r2 = x^2 + y^2,  rad = 1 + k1 r2 + k2 r2^2 + k3 r2^3
//   x'' = x rad + 2 p1 x y + p2 (r2 + 2 x^2),   y'' = y rad + p1 (r2 + 2 y^2) + 2 p2 x y,   u = fx x'' + cx,  v = fy y'' + cy
// One wavefront per view: every lane differentiates its board point's two residuals with respect to the 9 intrinsics and the view's 6 pose
// coordinates by FORWARD-MODE automatic differentiation (a value + 15 partials: the projection is written once, no hand-derived Jacobian
// to get wrong), the rows meet in LDS, and the lanes then sum the 15 x 15 Gauss-Newton block, the gradient and the cost over the points in
// a fixed order.  A few thousand residuals: nothing here is tuned -- it is the initialiser of the hot path, not the hot path.
#include <hip/hip_runtime.h>
#include "mcba_kernels.h"

namespace mcba {

namespace {
constexpr int NP = 15;  // fx fy cx cy k1 k2 p1 p2 k3 | w (3) t (3)
struct Dual {
  double v;
  double d[NP];
};
__device__ __forceinline__ Dual constant(double v) {
  Dual r;
  r.v = v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = 0.0;
  return r;
}
__device__ __forceinline__ Dual variable(double v, int k) {
  Dual r = constant(v);
  r.d[k] = 1.0;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = fma(a.v, b.d[i], a.d[i] * b.v);
  return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, double s) {
  Dual r;
  r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, double s) {
  Dual r = a;
  r.v += s;
  return r;
}
// f(a) with f' given: the chain rule
__device__ __forceinline__ Dual chain(const Dual& a, double f, double fp) {
  Dual r;
  r.v = f;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = fp * a.d[i];
  return r;
}
__device__ __forceinline__ Dual reciprocal(const Dual& a) {
  const double iv = 1.0 / a.v;
  return chain(a, iv, -iv * iv);
}
// sin(t) / t and (1 - cos t) / t^2 as functions of q = t^2, with their derivatives in q (series below t^2 = 1e-6: the reference's
// rodrigues() divides by 1 at t = 0, geometry.py:8-35 -- the same limit)
__device__ __forceinline__ void rot_coefficients(const Dual& q, Dual& a, Dual& b) {
  const double qq = q.v;
  double av, ad, bv, bd;
  if (qq < 1e-6) {
    av = 1.0 - qq / 6.0 + qq * qq / 120.0;
    ad = -1.0 / 6.0 + qq / 60.0;
    bv = 0.5 - qq / 24.0 + qq * qq / 720.0;
    bd = -1.0 / 24.0 + qq / 360.0;
  } else {
    const double t = sqrt(qq), s = sin(t), c = cos(t);
    av = s / t;
    ad = (t * c - s) / (2.0 * t * qq);
    bv = (1.0 - c) / qq;
    bd = (t * s - 2.0 * (1.0 - c)) / (2.0 * qq * qq);
  }
  a = chain(q, av, ad);
  b = chain(q, bv, bd);
}

// residuals (observed - predicted) of one board point and their 15 partials
__device__ __forceinline__ void point_rows(const double* __restrict__ intr9, const double* __restrict__ pose6, const double X[3], double ou, double ov, double* row /* [32]: ju[15] jv[15] ru rv */) {
  Dual K[9], w[3], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) K[i] = variable(intr9[i], i);
#pragma unroll
  for (int i = 0; i < 3; ++i) { w[i] = variable(pose6[i], 9 + i); t[i] = variable(pose6[3 + i], 12 + i); }
  const Dual q = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  Dual a, b;
  rot_coefficients(q, a, b);
  // R X = X + a (w x X) + b (w x (w x X))
  Dual c1[3] = {w[1] * X[2] - w[2] * X[1], w[2] * X[0] - w[0] * X[2], w[0] * X[1] - w[1] * X[0]};
  Dual c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};
  Dual Xc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Xc[i] = a * c1[i] + b * c2[i] + t[i] + X[i];
  const Dual iz = reciprocal(Xc[2]);
  const Dual x = Xc[0] * iz, y = Xc[1] * iz;
  const Dual xx = x * x, yy = y * y, xy = x * y;
  const Dual r2 = xx + yy;
  const Dual rad = (((K[8] * r2 + K[5]) * r2 + K[4]) * r2) + 1.0;
  const Dual xd = x * rad + K[6] * xy * 2.0 + K[7] * (r2 + xx * 2.0);
  const Dual yd = y * rad + K[6] * (r2 + yy * 2.0) + K[7] * xy * 2.0;
  const Dual u = K[0] * xd + K[2], v = K[1] * yd + K[3];
  const bool su = ou == ou, sv = ov == ov;   // (a missing scalar contributes nothing)
#pragma unroll
  for (int i = 0; i < NP; ++i) { row[i] = su ? -u.d[i] : 0.0; row[NP + i] = sv ? -v.d[i] : 0.0; }
  row[30] = su ? ou - u.v : 0.0;
  row[31] = sv ? ov - v.v : 0.0;
}
}  // namespace

// out[view][136] = upper triangle of sum_p (ju ju^T + jv jv^T) row by row (120) | gradient sum_p (ju ru + jv rv) (15) | 0.5 sum r^2
__global__ __launch_bounds__(64) void k_calib_views(const double2* __restrict__ uvs, const double* __restrict__ obj, const double* __restrict__ intr9, const double* __restrict__ poses, int N,
                                                    double* __restrict__ out) {
  __shared__ double rows[64][33];   // (33: the 64 lanes' rows start in different banks)
  const int view = blockIdx.x, lane = threadIdx.x;
  // entries of this lane: e = lane, lane + 64, lane + 128 (< 136)
  int ea[3], eb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = lane + 64 * k;
    ea[k] = -1; eb[k] = 0;
    if (e < 120) {
      int i = 0, rem = e;
      while (rem >= NP - i) { rem -= NP - i; ++i; }
      ea[k] = i; eb[k] = i + rem;
    } else if (e < 135) { ea[k] = e - 120; eb[k] = -1; }
    else if (e == 135) { ea[k] = 0; eb[k] = -2; }
  }
  double acc[3] = {0.0, 0.0, 0.0};
  for (int p0 = 0; p0 < N; p0 += 64) {
    const int p = p0 + lane;
    if (p < N) {
      const double2 o = uvs[(size_t)view * N + p];
      const double X[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
      double row[32];
      point_rows(intr9, poses + 6 * (size_t)view, X, o.x, o.y, row);
#pragma unroll
      for (int i = 0; i < 32; ++i) rows[lane][i] = row[i];
    }
    __syncthreads();
    const int np = min(64, N - p0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (ea[k] < 0) continue;
      double s = 0.0;
      if (eb[k] >= 0) for (int j = 0; j < np; ++j) s += rows[j][ea[k]] * rows[j][eb[k]] + rows[j][NP + ea[k]] * rows[j][NP + eb[k]];
      else if (eb[k] == -1) for (int j = 0; j < np; ++j) s += rows[j][ea[k]] * rows[j][30] + rows[j][NP + ea[k]] * rows[j][31];
      else for (int j = 0; j < np; ++j) s += 0.5 * (rows[j][30] * rows[j][30] + rows[j][31] * rows[j][31]);
      acc[k] += s;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (ea[k] >= 0) out[(size_t)view * 136 + lane + 64 * k] = acc[k];
}

void launch_calib_views(hipStream_t st, const double* uvs, const double* obj, const double* intr9, const double* poses, int V, int N, double* out) {
  k_calib_views<<<dim3(V), dim3(64), 0, st>>>(reinterpret_cast<const double2*>(uvs), obj, intr9, poses, N, out);
}

}  // namespace mcba
