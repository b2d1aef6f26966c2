// mcba_triangulate.hip -- robust multi-view triangulation (reference geometry.py:361-433).
//
// One LANE per 3-D point: undistort its detection in every camera (OpenCV's fixed-point iteration for the 5-coefficient
// model), triangulate it linearly from every camera pair that sees it (4x4 DLT system, right singular vector of the
// smallest singular value by one-sided Jacobi -- no normal equations, so nothing is squared), take the per-coordinate
// nan-median over the pairs.  Everything stays in registers; the only memory traffic is 16 B in per (camera, point) and
// 24 B out per point, the arithmetic (~3 k FP64 instructions per pair) makes the kernel FP64-VALU bound.
// The camera count is a template parameter so that the per-camera / per-pair arrays are register arrays.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mcba_kernels.h"
#include "mcba_math.h"

namespace mcba {

// right singular vector of the smallest singular value of the 4x4 matrix whose COLUMNS are a[0..3] (each a 4-vector)
__device__ __forceinline__ void null_vector4(double (&a)[4][4], double (&x)[4]) {
  double v[4][4];  // v[k] = column k of V
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 7; ++sweep) {  // quadratic convergence: 4-5 sweeps reach FP64 for a 4x4; fixed count, branch-free
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { alpha = fma(a[p][k], a[p][k], alpha); beta = fma(a[q][k], a[q][k], beta); gamma = fma(a[p][k], a[q][k], gamma); }
        const bool rot = gamma * gamma > 1e-32 * alpha * beta;  // already orthogonal to FP64: identity
        const double g = rot ? gamma : 1.0;
        const double zeta = (beta - alpha) * fast_rcp(2.0 * g);
        const double az = fabs(zeta);
        double t = fast_rcp(az + sqrt(fma(zeta, zeta, 1.0)));
        t = zeta < 0.0 ? -t : t;
        t = rot ? t : 0.0;
        const double c = fast_rsqrt(fma(t, t, 1.0)), s = c * t;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double ap = a[p][k], aq = a[q][k];
          a[p][k] = fma(c, ap, -(s * aq));
          a[q][k] = fma(s, ap, c * aq);
          const double vp = v[p][k], vq = v[q][k];
          v[p][k] = fma(c, vp, -(s * vq));
          v[q][k] = fma(s, vp, c * vq);
        }
      }
  }
  double best = 1e300;
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = 0.0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) nrm = fma(a[p][k], a[p][k], nrm);
    const bool take = nrm < best;
    best = take ? nrm : best;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = take ? v[p][k] : x[k];
  }
}

template <int C>
__global__ __launch_bounds__(256) void k_triangulate(const double2* __restrict__ uvs, const TriCams cams, double* __restrict__ out, size_t npts, int iters) {
  constexpr int NP = C * (C - 1) / 2;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts) return;
  double ux[C], uy[C];
  bool ok[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const double2 o = uvs[(size_t)c * npts + p];  // (C, P) pairs: consecutive lanes, consecutive points
    ok[c] = o.x == o.x && o.y == o.y;
    const double fx = cams.K[c][0], fy = cams.K[c][1], cx = cams.K[c][2], cy = cams.K[c][3];
    const double k1 = cams.dist[c][0], k2 = cams.dist[c][1], p1 = cams.dist[c][2], p2 = cams.dist[c][3], k3 = cams.dist[c][4];
    const double x0 = (o.x - cx) / fx, y0 = (o.y - cy) / fy;
    double x = x0, y = y0;
    for (int it = 0; it < iters; ++it) {
      const double r2 = fma(x, x, y * y);
      const double icdist = 1.0 / fma(fma(fma(k3, r2, k2), r2, k1), r2, 1.0);
      const double dx = fma(2.0 * p1 * x, y, p2 * fma(2.0 * x, x, r2));
      const double dy = fma(p1, fma(2.0 * y, y, r2), 2.0 * p2 * x * y);
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    ux[c] = fma(x, fx, cx);
    uy[c] = fma(y, fy, cy);
  }
  double X[NP], Y[NP], Z[NP];
  int n = 0;
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
      for (int j = i + 1; j < C; ++j, ++k) {
        double a[4][4];  // a[col][row]
#pragma unroll
        for (int col = 0; col < 4; ++col) {
          a[col][0] = fma(ux[i], cams.P[i][8 + col], -cams.P[i][col]);
          a[col][1] = fma(uy[i], cams.P[i][8 + col], -cams.P[i][4 + col]);
          a[col][2] = fma(ux[j], cams.P[j][8 + col], -cams.P[j][col]);
          a[col][3] = fma(uy[j], cams.P[j][8 + col], -cams.P[j][4 + col]);
        }
        const bool both = ok[i] && ok[j];
        if (!both) {  // keep the arithmetic finite; the result is discarded
#pragma unroll
          for (int col = 0; col < 4; ++col)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[col][r] = col == r ? 1.0 : 0.0;
        }
        double x[4];
        null_vector4(a, x);
        const double iw = 1.0 / x[3];
        const double big = 1e300;  // invalid pairs sort to the end
        X[k] = both ? x[0] * iw : big;
        Y[k] = both ? x[1] * iw : big;
        Z[k] = both ? x[2] * iw : big;
        n += both ? 1 : 0;
      }
  }
  // per-coordinate nan-median: sort (odd-even transposition network, NP passes), pick the middle (or the mean of two)
  auto median = [&](double (&v)[NP]) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
#pragma unroll
      for (int k = pass & 1; k + 1 < NP; k += 2) {
        const double lo = fmin(v[k], v[k + 1]), hi = fmax(v[k], v[k + 1]);
        v[k] = lo; v[k + 1] = hi;
      }
    double m0 = 0.0, m1 = 0.0;
    const int i0 = (n - 1) >> 1, i1 = n >> 1;
#pragma unroll
    for (int k = 0; k < NP; ++k) { m0 = k == i0 ? v[k] : m0; m1 = k == i1 ? v[k] : m1; }
    return n > 0 ? 0.5 * (m0 + m1) : __builtin_nan("");
  };
  const double mx = median(X), my = median(Y), mz = median(Z);
  out[3 * p] = mx; out[3 * p + 1] = my; out[3 * p + 2] = mz;
}

int launch_triangulate(hipStream_t st, int C, const double* uvs, const TriCams& cams, double* out, size_t npts, int iters) {
  const dim3 grid((unsigned)((npts + 255) / 256)), block(256);
  const double2* u2 = reinterpret_cast<const double2*>(uvs);
  switch (C) {
    case 2: k_triangulate<2><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    case 3: k_triangulate<3><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    case 4: k_triangulate<4><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    case 5: k_triangulate<5><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    case 6: k_triangulate<6><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    case 7: k_triangulate<7><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    case 8: k_triangulate<8><<<grid, block, 0, st>>>(u2, cams, out, npts, iters); break;
    default: return 1;
  }
  return 0;
}

}  // namespace mcba
