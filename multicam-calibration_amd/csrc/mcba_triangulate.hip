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
#include <algorithm>
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
        // (a pair whose null vector has x[3] == 0 -- a point at infinity for that pair -- gives NaN or +-inf: such a pair does not
        // count at all, in any coordinate; np.nanmedian drops the NaNs per coordinate, geometry.py:432)
        const double vx = x[0] * iw, vy = x[1] * iw, vz = x[2] * iw;
        const bool keep = both && fabs(vx) < big && fabs(vy) < big && fabs(vz) < big;
        X[k] = keep ? vx : big;
        Y[k] = keep ? vy : big;
        Z[k] = keep ? vz : big;
        n += keep ? 1 : 0;
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

// Any camera count (9 .. 64): ONE WAVEFRONT per 3-D point, the camera pairs spread over its lanes.  Lanes c < C undistort
// the detection of camera c (LDS), every lane triangulates the pairs k = lane, lane + 64, ... (same 4x4 null vector as
// above) into LDS, and the per-coordinate nan-median over the <= C (C - 1) / 2 pair results is found by rank counting: each
// lane counts, for its own values, how many values precede them (ties broken by index) -- the values of rank (n-1)/2 and
// n/2 are the two middle ones.  Camera constants come from global memory (TriCam per camera) instead of kernel arguments.
struct TriCam {
  double P[12], K[4], dist[5];
};

__global__ __launch_bounds__(256) void k_triangulate_wave(const double2* __restrict__ uvs, const TriCam* __restrict__ cams, double* __restrict__ out, size_t npts, int C, int NP, int iters, int ppb) {
  extern __shared__ double s_tri[];  // per point slot: ux[C] uy[C] ok[C] | X[NP] Y[NP] Z[NP] | m[2]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= ppb) return;
  const size_t p = (size_t)blockIdx.x * ppb + wave;
  if (p >= npts) return;  // whole wavefront
  const int slot = 3 * C + 3 * NP + 2;
  double* ux = s_tri + (size_t)wave * slot;
  double* uy = ux + C;
  double* okf = uy + C;
  double* XYZ = okf + C;
  double* mm = XYZ + 3 * NP;
  for (int c = lane; c < C; c += 64) {
    const double2 o = uvs[(size_t)c * npts + p];
    const TriCam& cm = cams[c];
    double x0 = (o.x - cm.K[2]) / cm.K[0], y0 = (o.y - cm.K[3]) / cm.K[1];
    double x = x0, y = y0;
    for (int it = 0; it < iters; ++it) {
      const double r2 = fma(x, x, y * y);
      const double icdist = 1.0 / fma(fma(fma(cm.dist[4], r2, cm.dist[1]), r2, cm.dist[0]), r2, 1.0);
      const double dx = fma(2.0 * cm.dist[2] * x, y, cm.dist[3] * fma(2.0 * x, x, r2));
      const double dy = fma(cm.dist[2], fma(2.0 * y, y, r2), 2.0 * cm.dist[3] * x * y);
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    ux[c] = fma(x, cm.K[0], cm.K[2]);
    uy[c] = fma(y, cm.K[1], cm.K[3]);
    okf[c] = (o.x == o.x && o.y == o.y) ? 1.0 : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const double big = 1e300;  // invalid pairs sort to the end
  int nloc = 0;
  for (int k = lane; k < NP; k += 64) {
    // pair k -> (i, j), i < j, in the order (0,1), (0,2), ..., (0,C-1), (1,2), ...
    int i = 0, rem = k;
    while (rem >= C - 1 - i) { rem -= C - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const TriCam& ci = cams[i];
    const TriCam& cj = cams[j];
    double a[4][4];
#pragma unroll
    for (int col = 0; col < 4; ++col) {
      a[col][0] = fma(ux[i], ci.P[8 + col], -ci.P[col]);
      a[col][1] = fma(uy[i], ci.P[8 + col], -ci.P[4 + col]);
      a[col][2] = fma(ux[j], cj.P[8 + col], -cj.P[col]);
      a[col][3] = fma(uy[j], cj.P[8 + col], -cj.P[4 + col]);
    }
    const bool both = okf[i] != 0.0 && okf[j] != 0.0;
    if (!both) {
#pragma unroll
      for (int col = 0; col < 4; ++col)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[col][r] = col == r ? 1.0 : 0.0;
    }
    double xh[4];
    null_vector4(a, xh);
    const double iw = 1.0 / xh[3];
    const double vx = xh[0] * iw, vy = xh[1] * iw, vz = xh[2] * iw;
    const bool keep = both && fabs(vx) < big && fabs(vy) < big && fabs(vz) < big;   // (see k_triangulate: NaN / infinite results do not count)
    XYZ[k] = keep ? vx : big;
    XYZ[NP + k] = keep ? vy : big;
    XYZ[2 * NP + k] = keep ? vz : big;
    nloc += keep ? 1 : 0;
  }
  int n = nloc;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int i0 = (n - 1) >> 1, i1 = n >> 1;
  double res[3];
  for (int d = 0; d < 3; ++d) {
    const double* v = XYZ + d * NP;
    if (lane < 2) mm[lane] = __builtin_nan("");   // (never read stale: the two middle ranks exist whenever n > 0)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int k = lane; k < NP; k += 64) {
      const double mine = v[k];
      int rank = 0;
      for (int m = 0; m < NP; ++m) {
        const double o = v[m];
        rank += (o < mine || (o == mine && m < k)) ? 1 : 0;
      }
      if (rank == i0) mm[0] = mine;
      if (rank == i1) mm[1] = mine;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    res[d] = n > 0 ? 0.5 * (mm[0] + mm[1]) : __builtin_nan("");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) { out[3 * p] = res[0]; out[3 * p + 1] = res[1]; out[3 * p + 2] = res[2]; }
}

// cams_dev: C x TriCam in device memory (same content as TriCams, any camera count)
int launch_triangulate_wave(hipStream_t st, int C, const double* uvs, const void* cams_dev, double* out, size_t npts, int iters) {
  const int NP = C * (C - 1) / 2;
  const size_t slot = ((size_t)3 * C + 3 * NP + 2) * sizeof(double);
  int ppb = (int)std::min<size_t>(4, (64 * 1024) / slot);
  if (ppb < 1) return 1;
  const dim3 grid((unsigned)((npts + ppb - 1) / ppb)), block(256);
  k_triangulate_wave<<<grid, block, slot * ppb, st>>>(reinterpret_cast<const double2*>(uvs), static_cast<const TriCam*>(cams_dev), out, npts, C, NP, iters, ppb);
  return 0;
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
