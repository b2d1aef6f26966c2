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

#include <cstdlib>

#include "mcba_device.h"
#include "mcba_kernels.h"
#include "mcba_math.h"
#include "mcba_pnp_math.h"   // the per-view arithmetic (shared with the host harness tests/hostcheck/hostcheck.cpp)

namespace mcba {

namespace {

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

// a view's points dealt out to four neighbouring lanes (mcba_pnp_math.h: WholeView is the one-lane form): partial sums meet by two quad_perm
// DPP steps, after which the four lanes hold the same bits.  Every loop of k_pnp is wave-uniform, so the four lanes are always active together.
struct QuadView {
  static constexpr int parts = 4;
  int lane;
  __device__ __forceinline__ int part() const { return lane & 3; }
  __device__ __forceinline__ double sum(double v) const {
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    return dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  }
  __device__ __forceinline__ bool all(bool b) const { return ((__ballot(b) >> (lane & ~3)) & 0xFull) == 0xFull; }
};
template <int PARTS> struct ViewSplit;
template <> struct ViewSplit<1> { using type = WholeView; __device__ static WholeView make(int) { return WholeView{}; } };
template <> struct ViewSplit<4> { using type = QuadView; __device__ static QuadView make(int lane) { return QuadView{lane}; } };

// PARTS = lanes per view (1 or 4): 64 / PARTS views per wavefront.  The per-view dependent chain is what a launch takes, and the point loops are
// nine tenths of it: four lanes per view cut them in four (which form runs: pnp_lanes_per_view below).
template <int MODE, bool DENSE, int PARTS>
__global__ __launch_bounds__(64) void k_pnp(PnpArgs a) {
  constexpr int VPW = 64 / PARTS;
  const int lane = threadIdx.x;
  const auto split = ViewSplit<PARTS>::make(lane);
  const bool writer = lane % PARTS == 0;
  int c, f, vi;
  bool in_range;
  if (DENSE) {
    c = blockIdx.y; f = blockIdx.x * VPW + lane / PARTS; vi = 0;
    in_range = f < a.F;
  } else {
    vi = blockIdx.x * VPW + lane / PARTS;
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
  // ---- pass 1: complete?  Hartley normalisation of the image points;  pass 2: the DLT's normal matrix and its block Cholesky factor
  bool complete;
  double mx, my, ss;
  view_normalisation(image_point, N, in_range, split, complete, mx, my, ss);
  DltFactor dlt;
  view_dlt_factor(image_point, a.obj, N, a.bmx, a.bmy, a.bs, complete, mx, my, ss, split, dlt);
  // inverse iteration for the smallest eigenvector, until no lane of the wavefront moves any more
  double h[9];
  dlt_start_vector(h);
  for (int it = 0; it < 60; ++it) {
    const double diff = dlt_inverse_iteration(dlt, h);
    const bool more = complete && !(diff <= 4e-16);
    if (!__any(more)) break;
  }
  double H[9];
  homography_denormalise(h, a.bmx, a.bmy, a.bs, mx, my, ss, H);
  const double nan = __builtin_nan("");
  if (MODE == MODE_HOMOGRAPHY) {
    bool ok = complete;
#pragma unroll
    for (int i = 0; i < 9; ++i) ok = ok && pnp_finite(H[i]);
    if (in_range && writer) {
#pragma unroll
      for (int i = 0; i < 9; ++i) a.out[(size_t)vi * 9 + i] = ok ? H[i] : nan;
      if (a.valid) a.valid[vi] = ok ? 1 : 0;
    }
    return;
  }

  // ---- pose from the homography, then Levenberg-Marquardt on the reprojection error in pixels, one problem per lane, until no lane is left
  double pose0[6];
  pose_from_homography(H, pose0);
  ViewLM lm;
  view_lm_init(lm, pose0, complete);
  auto observation = [&](int p, double& u, double& v) {
    const double2 o = op[(size_t)p * Fpad];
    u = o.x; v = o.y;
  };
  for (int it = 0; it < a.lm_iters; ++it) {
    double Hn[21], gn[6], cn;
    view_linearise(lm.trial, cam, a.obj, N, observation, complete, split, Hn, gn, cn);
    view_lm_decide(lm, Hn, gn, cn);
    if (!__any(!lm.done)) break;
    view_lm_step(lm);
  }
  const bool failed = lm.failed;
  const int evals = lm.evals;
  double pose[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pose[i] = lm.pose[i];
  bool ok = complete && !failed;
#pragma unroll
  for (int i = 0; i < 6; ++i) ok = ok && pnp_finite(pose[i]);
  if (!writer) return;
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

// Zhang's closed form (mcba_pnp_math.h) for every camera of a sampled view list: block = camera, the lanes share out the list and add their
// views' rows to the 6 x 6 normal matrix, lane 63 holds the wave's sums and solves.  intr9 [C][9] <- fx fy cx cy 0 0 0 0 0 (the start of the
// per-view poses and of the joint refinement); closed [C] <- 1 where the closed form was used, 0 for the fallback.
__global__ __launch_bounds__(64) void k_zhang(const double* __restrict__ H, const unsigned char* __restrict__ ok, const int* __restrict__ views, int nviews, const double* __restrict__ sizes,
                                              double* __restrict__ intr9, unsigned char* __restrict__ closed) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const double w = sizes[2 * c], h = sizes[2 * c + 1];
  const double s0 = w > h ? w : h, ox = 0.5 * (w - 1.0), oy = 0.5 * (h - 1.0), is0 = 1.0 / s0;
  double M[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) M[i] = 0.0;
  double n = 0.0;
  for (int i = lane; i < nviews; i += 64) {
    if (views[2 * i] != c || !ok[i]) continue;
    double Hv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hv[k] = H[(size_t)9 * i + k];
    zhang_accumulate(Hv, ox, oy, is0, M);
    n += 1.0;
  }
#pragma unroll
  for (int i = 0; i < 21; ++i) M[i] = wave_sum63(M[i]);
  n = wave_sum63(n);
  if (lane != 63) return;
  double K4[4];
  const bool used = zhang_solve(M, (int)n, w, h, K4);
#pragma unroll
  for (int i = 0; i < 9; ++i) intr9[9 * c + i] = i < 4 ? K4[i] : 0.0;
  if (closed) closed[c] = used ? 1 : 0;
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

// calibration.py:226-235 on the device: the medians of the tree's pairwise transforms (the two middle order statistics the radix select left in
// its states: word 2 = count, word 3 = the value's bits, `stride` words per state, two states per component) chained from the root --
// T_world->c2 = T_c1->c2 T_world->c1, edges ordered so that c1 is placed before c2 (the caller checks) -- into ext [C][6]; transforms [E][6] and
// counts [E] (frames the pair shares) for the caller.  One wavefront; only the C - 1 products of the chain itself run on one lane.
__global__ __launch_bounds__(64) void k_pose_chain(const unsigned long long* __restrict__ sel, int stride, const int* __restrict__ edges, int n_edges, int root, int C,
                                                   double* __restrict__ ext, double* __restrict__ transforms, double* __restrict__ counts) {
  __shared__ double Tm[40][12];   // world -> camera: R (9, row-major), t (3)
  __shared__ double Te[39][12];   // the edges' transforms as matrices
  __shared__ double med[39 * 6];
  const int lane = threadIdx.x;
  const double nan = __builtin_nan("");
  // the medians: lane = (edge, component) -- every lane's loads in flight together, not 12 E round trips of one lane
  for (int i = lane; i < 6 * n_edges; i += 64) {
    const unsigned long long cnt = sel[(size_t)(2 * i) * stride + 2];
    const unsigned long long a = sel[(size_t)(2 * i) * stride + 3], b = sel[(size_t)(2 * i + 1) * stride + 3];
    const double m = cnt ? 0.5 * (__longlong_as_double((long long)a) + __longlong_as_double((long long)b)) : nan;   // np.median: the mean of the two middle values
    med[i] = m;
    transforms[i] = m;
    if (i % 6 == 0) counts[i / 6] = (double)cnt;
  }
  __syncthreads();
  for (int e = lane; e < n_edges; e += 64) {   // lane = edge: its rotation matrix
    double t[3] = {med[6 * e], med[6 * e + 1], med[6 * e + 2]}, Re[9];
    rot_only(t, Re);
#pragma unroll
    for (int k = 0; k < 9; ++k) Te[e][k] = Re[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Te[e][9 + k] = med[6 * e + 3 + k];
  }
  for (int c = lane; c < C; c += 64)
#pragma unroll
    for (int k = 0; k < 12; ++k) Tm[c][k] = c == root ? (k == 0 || k == 4 || k == 8 ? 1.0 : 0.0) : nan;
  __syncthreads();
  if (lane == 0) {   // the chain itself is sequential: C - 1 products
    for (int e = 0; e < n_edges; ++e) {
      const int c1 = edges[2 * e], c2 = edges[2 * e + 1];
      double Re[9], R1[9], t1[3], R2[9], t2[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) { Re[k] = Te[e][k]; R1[k] = Tm[c1][k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) t1[k] = Tm[c1][9 + k];
      mm33(Re, R1, R2);
      mv3(Re, t1, t2);
#pragma unroll
      for (int k = 0; k < 9; ++k) Tm[c2][k] = R2[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Tm[c2][9 + k] = t2[k] + Te[e][9 + k];
    }
  }
  __syncthreads();
  for (int c = lane; c < C; c += 64) {   // lane = camera: rotation vector (the root's: exactly 0)
    double R[9], w[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Tm[c][k];
    rotvec_from_matrix(R, w);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ext[6 * c + k] = w[k];
      ext[6 * c + 3 + k] = Tm[c][9 + k];
    }
  }
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

// Lanes per view.  Measured on one box (profiles/round6/NOTES_round6.md 1.2): four lanes win wherever the serial chain per view counts -- 0.67 ->
// 0.32 ms for the tutorial's 12 780 views, 2.11 -> 1.77 ms at 240 000 views of 54 points -- and lose where the launch is a stream of point
// loops over more data than the caches hold (300 000 views of 200 points: 5.0 -> 6.3 ms; a quad's loads are 256-byte pieces of four rows).
// MCBA_PNP_LANES = 1 | 4 overrides (read per launch: the tests run both forms in one process).
int pnp_lanes_per_view(long views, int N) {
  const char* e = getenv("MCBA_PNP_LANES");
  const int forced = e ? atoi(e) : 0;
  if (forced == 1 || forced == 4) return forced;
  return (double)views * N <= 2.4e7 ? 4 : 1;
}

void launch_pnp(hipStream_t st, int mode, const double* obs_t, const double* obj, const double* intr9, const int* views, int nviews, const double* bn3, int C, int F, int N, int Fpad, int und_iters, int lm_iters,
                double* out, double* poses_t, unsigned char* valid, unsigned char* nit) {
  PnpArgs a;
  a.obs_t = reinterpret_cast<const double2*>(obs_t); a.obj = obj; a.intr9 = intr9; a.views = views;
  a.bmx = bn3[0]; a.bmy = bn3[1]; a.bs = bn3[2];
  a.nviews = nviews; a.C = C; a.F = F; a.N = N; a.Fpad = Fpad; a.und_iters = und_iters; a.lm_iters = lm_iters;
  a.out = out; a.poses_t = poses_t; a.valid = valid; a.nit = nit;
  const long total = views ? (long)nviews : (long)C * F;
  const bool quad = pnp_lanes_per_view(total, N) == 4;
  if (views) {
    const dim3 grid1((nviews + 63) / 64), grid4((nviews + 15) / 16);
    if (mode == MODE_HOMOGRAPHY) {
      if (quad) k_pnp<MODE_HOMOGRAPHY, false, 4><<<grid4, dim3(64), 0, st>>>(a);
      else k_pnp<MODE_HOMOGRAPHY, false, 1><<<grid1, dim3(64), 0, st>>>(a);
    } else {
      if (quad) k_pnp<MODE_POSE, false, 4><<<grid4, dim3(64), 0, st>>>(a);
      else k_pnp<MODE_POSE, false, 1><<<grid1, dim3(64), 0, st>>>(a);
    }
  } else {
    if (quad) k_pnp<MODE_POSE, true, 4><<<dim3(Fpad / 16, C), dim3(64), 0, st>>>(a);
    else k_pnp<MODE_POSE, true, 1><<<dim3(Fpad / 64, C), dim3(64), 0, st>>>(a);
  }
}

void launch_zhang(hipStream_t st, const double* H, const unsigned char* ok, const int* views, int nviews, const double* sizes, int C, double* intr9, unsigned char* closed) {
  k_zhang<<<dim3(C), dim3(64), 0, st>>>(H, ok, views, nviews, sizes, intr9, closed);
}

void launch_pose_pairs(hipStream_t st, const double* poses, size_t sc, size_t sf, size_t sk, const int* edges, int n_edges, int F, int Fpad, double* rel) {
  k_pose_pairs<<<dim3(Fpad / 64, n_edges), dim3(64), 0, st>>>(PoseView{poses, sc, sf, sk}, edges, F, Fpad, rel);
}

void launch_pose_chain(hipStream_t st, const void* sel, size_t sel_state_bytes, const int* edges, int n_edges, int root, int C, double* ext, double* transforms, double* counts) {
  k_pose_chain<<<dim3(1), dim3(64), 0, st>>>(static_cast<const unsigned long long*>(sel), (int)(sel_state_bytes / 8), edges, n_edges, root, C, ext, transforms, counts);
}

void launch_pose_consensus(hipStream_t st, const double* poses, size_t sc, size_t sf, size_t sk, const double* ext, int C, int F, int Fpad, double* world, double* out) {
  k_pose_consensus<<<dim3(Fpad / 64), dim3(64), 0, st>>>(PoseView{poses, sc, sf, sk}, ext, C, F, Fpad, world, out);
}

}  // namespace mcba
