// mcba_kernels.hip -- gfx950 kernels of the bundle-adjustment hot path (FP64, no MFMA: HBM/VALU bound).
//
// Work decomposition:
//   k_gram / k_cost     one LANE per (camera c, frame f): lanes of a wavefront are 64 consecutive frames of
//                       one camera, the loop runs over the board points.  Observations are stored
//                       [camera][point][frame] so each iteration's load is one coalesced 1 KiB line group;
//                       all accumulation (12x12 local Gram matrix) is lane-local -- no shuffles in the loop.
//                       Camera intrinsics + pose are staged in LDS once per workgroup.
//   k_frame_factor      one lane per frame: 6x6 Cholesky of the damped frame block, z = L^-1 g_f.
//   k_syrk              Y = W L^-T built in LDS from the records, S -= Y Y^T register-tiled over 12x12 camera-block pairs.
//   k_reduce_system     fixed-order second-stage reduction (deterministic; no FP64 atomics anywhere).
//   k_backsub           one lane per frame: frame steps, trial parameters, predicted-reduction terms.
//   k_jacobian          one wavefront per (camera, frame), one lane per board point; rows transposed through
//                       LDS so the 288 B/observation Jacobian blocks leave as coalesced 16 B/lane stores.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mcba_math.h"
#include "mcba_kernels.h"

namespace mcba {

// ---------------------------------------------------------------- small device helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Sum over the 64 lanes, result valid in lane 63 only.  DPP moves (pure VALU, no LDS round trip):
// xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror -> every lane holds its 16-lane row total;
// row_bcast15 (rows 1,3) and row_bcast31 (rows 2,3) fold the four rows into row 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, ROW_MASK, 0xF, false);
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, ROW_MASK, 0xF, false);
  return v + b.d;
}
__device__ __forceinline__ double wave_sum63(double v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
// value known to be wave-uniform -> SGPR pair (frees two VGPRs and a ds_read per use)
__device__ __forceinline__ double uni(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
__device__ __forceinline__ bool is_num(double v) { return v == v; }

template <int LOSS>
__device__ __forceinline__ void obs_weights(double r, bool valid, double fs2, double ifs2, double& cost, double& w2, double& g) {
  double rh, gw, ww;
  loss_weights<LOSS>(r, fs2, ifs2, rh, gw, ww);
  cost += valid ? rh : 0.0;
  w2 = valid ? lm_weight(gw, ww) : 0.0;
  g = valid ? gw * r : 0.0;
}

// ---------------------------------------------------------------- observation re-layout
// raw (C,F,N,2) -> obs_t [C][N][Fpad] (u,v); frames f >= F are NaN (= missing, contribute nothing)
__global__ void k_transpose_obs(const double2* __restrict__ raw, double2* __restrict__ obs_t, int C, int F, int N, int Fpad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)C * N * Fpad;
  if (i >= total) return;
  int f = (int)(i % Fpad);
  size_t cp = i / Fpad;
  int p = (int)(cp % N);
  int c = (int)(cp / N);
  double2 v;
  v.x = v.y = __builtin_nan("");
  if (f < F) v = raw[((size_t)c * F + f) * N + p];
  obs_t[i] = v;
}

// ---------------------------------------------------------------- k_gram: linearise
// grid (ceil(nfb/4), C, 2), block 256 = 4 wavefronts = 4 frame blocks of one camera; blockIdx.z is the ROLE
// (mcba_math.h: role A = [A|P] block -> V, g_f, W rows of rho/t, U_(rho,t)x(rho,t); role B = intrinsics blocks).
// Splitting the 87 accumulators over two wavefronts keeps each under 256 VGPRs, so two waves share a SIMD and
// hide each other's FP64 / memory latency; the roles never exchange data.
template <int LOSS, int ROLE>
__device__ __forceinline__ void gram_body(const CamConst& s_cam, const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x,
                                          double* __restrict__ rec, double* __restrict__ gpart, int c, int fb, int lane, int C, int N, int Fpad, int nfb, double fs2, double ifs2) {
  const int f = fb * 64 + lane;
  Intr K;
  K.fx = uni(s_cam.fx); K.fy = uni(s_cam.fy); K.cx = uni(s_cam.cx); K.cy = uni(s_cam.cy); K.k1 = uni(s_cam.k1); K.k2 = uni(s_cam.k2);
  double Rc[9], tc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = uni(s_cam.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) tc[i] = uni(s_cam.t[i]);

  const double* pose = x + 12 * C + 6 * (size_t)f;
  PairConst pc;
  {
    double pz[6], Rf[9];
#pragma unroll
    for (int i = 0; i < 6; ++i) pz[i] = pose[i];
    rot_only(pz, Rf);
    make_pair_const(Rc, tc, Rf, pz + 3, pc);
  }

  constexpr bool DO_A = ROLE != 1, DO_B = ROLE != 0;
  GramA ga;
  GramB gb;
  if (DO_A) gram_zero(ga);
  if (DO_B) gram_zero(gb);
  double cost = 0.0;
  bool any = false;
  const double2* op = obs_t + (size_t)c * N * Fpad + f;
  double2 o_next = op[0];
  double xn0 = obj[0], xn1 = obj[1], xn2 = obj[2];  // wave-uniform (scalar) loads, prefetched one point ahead
  for (int p = 0; p < N; ++p) {
    double2 o2 = o_next;
    double Xo[3] = {xn0, xn1, xn2};
    if (p + 1 < N) {
      o_next = op[(size_t)(p + 1) * Fpad];
      xn0 = obj[3 * p + 3]; xn1 = obj[3 * p + 4]; xn2 = obj[3 * p + 5];
    }
    bool vu = is_num(o2.x), vv = is_num(o2.y);
    if (vu || vv) {
      any = true;
      ObsRows o;
      obs_rows_t<DO_B>(K, pc, Xo, o);
      double wu2, wv2, gu, gv;
      obs_weights<LOSS>(o2.x - o.up, vu, fs2, ifs2, cost, wu2, gu);
      obs_weights<LOSS>(o2.y - o.vp, vv, fs2, ifs2, cost, wv2, gv);
      if (DO_A) gram_add(ga, o, wu2, wv2, gu, gv);
      if (DO_B) gram_add(gb, o, wu2, wv2, gu, gv);
    }
  }

  // ---- expand once per (c,f): this role's part of W, V, g_f (record) and of U, g_c (reduced over the wave)
  ChainConst ch;
  {
    double pz[6], Rf[9], Jrf[9], Jrc[9];
#pragma unroll
    for (int i = 0; i < 6; ++i) pz[i] = pose[i];
    rot_and_jr(pz, Rf, Jrf);
#pragma unroll
    for (int i = 0; i < 9; ++i) Jrc[i] = uni(s_cam.Jr[i]);
    make_chain_const(Rc, Jrc, Rf, Jrf, pz + 3, ch);
  }
  // records are wave tiles rec[camera][frame block][k = 0..99][lane]: every store below is 512 contiguous bytes
  double* r = rec + ((size_t)c * nfb + fb) * (MCBA_REC * 64) + lane;
  double* gp = gpart + ((size_t)c * nfb + fb) * MCBA_GP;
  const bool writer = lane == 63;
  if constexpr (DO_A) {
    double U[78], gc[12], W[72], V[21], gf[6];
    gram_expand(ga, ch, U, gc, W, V, gf);
#pragma unroll
    for (int i = 36; i < 72; ++i) r[i * 64] = W[i];
#pragma unroll
    for (int i = 0; i < 21; ++i) r[(72 + i) * 64] = V[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) r[(93 + i) * 64] = gf[i];
#pragma unroll
    for (int a = 6; a < 12; ++a)
#pragma unroll
      for (int b = a; b < 12; ++b) {
        double sm = wave_sum63(U[tri12(a, b)]);
        if (writer) gp[tri12(a, b)] = sm;
      }
#pragma unroll
    for (int a = 6; a < 12; ++a) {
      double sm = wave_sum63(gc[a]);
      if (writer) gp[78 + a] = sm;
    }
    double cs = wave_sum63(cost);
    double nv = wave_sum63(any ? 1.0 : 0.0);
    if (writer) { gp[90] = cs; gp[91] = nv; }
  }
  if constexpr (DO_B) {
    double U[78], gc[12], W[72];
    gram_expand(gb, ch, U, gc, W);
#pragma unroll
    for (int i = 0; i < 36; ++i) r[i * 64] = W[i];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 12; ++b) {
        double sm = wave_sum63(U[tri12(a, b)]);
        if (writer) gp[tri12(a, b)] = sm;
      }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double sm = wave_sum63(gc[a]);
      if (writer) gp[78 + a] = sm;
    }
  }
}

// Split roles: grid.z = 2, <= 256 VGPRs, two waves per SIMD.
template <int LOSS>
__global__ __launch_bounds__(256, 2) void k_gram_split(const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x,
                                                       double* __restrict__ rec, double* __restrict__ gpart, int C, int N, int Fpad, int nfb, double fs2, double ifs2) {
  __shared__ CamConst s_cam;
  const int c = blockIdx.y;
  if (threadIdx.x == 0) make_cam_const(x + 12 * c, s_cam);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fb = blockIdx.x * 4 + wave;
  if (fb >= nfb) return;
  if (blockIdx.z == 0) gram_body<LOSS, 0>(s_cam, obs_t, obj, x, rec, gpart, c, fb, lane, C, N, Fpad, nfb, fs2, ifs2);
  else gram_body<LOSS, 1>(s_cam, obs_t, obj, x, rec, gpart, c, fb, lane, C, N, Fpad, nfb, fs2, ifs2);
}

// Both roles in one lane: grid.z = 1, one wave per SIMD (all 87 accumulators + temporaries in the 512-register file).
template <int LOSS>
__global__ __launch_bounds__(256) void k_gram(const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x,
                                                 double* __restrict__ rec, double* __restrict__ gpart, int C, int N, int Fpad, int nfb, double fs2, double ifs2) {
  __shared__ CamConst s_cam;  // camera intrinsics + pose (R, t, Jr) staged once per workgroup
  const int c = blockIdx.y;
  if (threadIdx.x == 0) make_cam_const(x + 12 * c, s_cam);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fb = blockIdx.x * 4 + wave;
  if (fb >= nfb) return;
  gram_body<LOSS, 2>(s_cam, obs_t, obj, x, rec, gpart, c, fb, lane, C, N, Fpad, nfb, fs2, ifs2);
}

// ---------------------------------------------------------------- k_cost: robust cost only (trial points), optional residual vector
// grid (ceil(nfb/4), C, nch): like k_gram, but the board points are split into nch chunks (blockIdx.z) so that
// ~4 wavefronts per SIMD are in flight -- the loop body is short and latency-bound at one wave per SIMD.
template <int LOSS, bool WRITE_RES>
__global__ __launch_bounds__(256) void k_cost(const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x,
                                              double* __restrict__ cpart, double* __restrict__ res, int C, int F, int N, int Fpad, int nfb, int nch, double fs2, double ifs2) {
  __shared__ CamConst s_cam;
  const int c = blockIdx.y;
  if (threadIdx.x == 0) make_cam_const(x + 12 * c, s_cam);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fb = blockIdx.x * 4 + wave;
  if (fb >= nfb) return;
  const int f = fb * 64 + lane;
  const int ch = blockIdx.z;
  const int p0 = (int)(((long long)N * ch) / nch), p1 = (int)(((long long)N * (ch + 1)) / nch);
  Intr K;
  K.fx = uni(s_cam.fx); K.fy = uni(s_cam.fy); K.cx = uni(s_cam.cx); K.cy = uni(s_cam.cy); K.k1 = uni(s_cam.k1); K.k2 = uni(s_cam.k2);
  double Rc[9], tc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = uni(s_cam.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) tc[i] = uni(s_cam.t[i]);
  const double* pose = x + 12 * C + 6 * (size_t)f;
  double pz[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pz[i] = pose[i];
  PairConst pc;
  {
    double Rf[9];
    rot_only(pz, Rf);
    make_pair_const(Rc, tc, Rf, pz + 3, pc);
  }
  double cost = 0.0, nres = 0.0;
  const double2* op = obs_t + (size_t)c * N * Fpad + f;
  double2 o_next = op[(size_t)p0 * Fpad];
  for (int p = p0; p < p1; ++p) {
    double2 o2 = o_next;
    if (p + 1 < p1) o_next = op[(size_t)(p + 1) * Fpad];
    bool vu = is_num(o2.x), vv = is_num(o2.y);
    double ru = 0.0, rv = 0.0;
    if (vu || vv) {
      double Xo[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
      double up, vp;
      project_only(K, pc, Xo, up, vp);
      double rh, gw, w2;
      ru = o2.x - up; rv = o2.y - vp;
      loss_weights<LOSS>(ru, fs2, ifs2, rh, gw, w2);
      cost += vu ? rh : 0.0;
      loss_weights<LOSS>(rv, fs2, ifs2, rh, gw, w2);
      cost += vv ? rh : 0.0;
      nres += (vu ? 1.0 : 0.0) + (vv ? 1.0 : 0.0);
    }
    if (WRITE_RES && f < F) {
      // (C,F,N,2) order of the reference's residual vector before NaN removal
      *reinterpret_cast<double2*>(res + (((size_t)c * F + f) * N + p) * 2) = make_double2(vu ? ru : 0.0, vv ? rv : 0.0);
    }
  }
  double cs = wave_sum(cost), ns = wave_sum(nres);
  size_t o = 2 * (((size_t)c * nfb + fb) * nch + ch);
  if (lane == 0) { cpart[o] = cs; cpart[o + 1] = ns; }
}

// ---------------------------------------------------------------- k_frame_factor
// lane = frame.  V_f = sum_c V_cf; D_f = diag(V_f) (Marquardt); L L^T = V_f + lambda D_f; z = L^-1 g_f.
// fbuf[f] = {L(21), z(6), g_f(6), D_f(6), pad}.  Per-block partials: max |g_f|, #failed factorisations.
__global__ __launch_bounds__(256) void k_frame_factor(const double* __restrict__ rec, double* __restrict__ fbuf, double* __restrict__ fpart, int C, int F, int Fpad, double lambda) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  double gmax = 0.0, nfail = 0.0;
  if (f < F) {
    double V[21], gf[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) V[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) gf[k] = 0.0;
    const int nfb = Fpad >> 6;
    for (int cc = 0; cc < C; ++cc) {
      const double* r = rec + ((size_t)cc * nfb + (f >> 6)) * (MCBA_REC * 64) + 72 * 64 + (f & 63);
      double t[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) t[k] = r[k * 64];  // 27 coalesced loads in flight
#pragma unroll
      for (int k = 0; k < 21; ++k) V[k] += t[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) gf[k] += t[21 + k];
    }
    double D[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double d = V[tri6(k, k)];
      D[k] = d > 0.0 ? d : 1.0;
      V[tri6(k, k)] = d + lambda * D[k];
    }
    double Lp[21], id[6], z[6];
    bool ok = chol6(V, Lp);
#pragma unroll
    for (int k = 0; k < 6; ++k) id[k] = 1.0 / Lp[k * (k + 1) / 2 + k];
    fwd6(Lp, id, gf, z);
    double o[40];
#pragma unroll
    for (int k = 0; k < 21; ++k) o[k] = Lp[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { o[21 + k] = z[k]; o[27 + k] = gf[k]; o[33 + k] = D[k]; gmax = fmax(gmax, fabs(gf[k])); }
    o[39] = 0.0;
    double* fbp = fbuf + (size_t)f * MCBA_FB;
#pragma unroll
    for (int k = 0; k < 40; k += 2) *reinterpret_cast<double2*>(fbp + k) = make_double2(o[k], o[k + 1]);
    nfail = ok ? 0.0 : 1.0;
  }
  __shared__ double s_m[4], s_n[4];
  double wm = wave_max(gmax), wn = wave_sum(nfail);
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = wm; s_n[threadIdx.x >> 6] = wn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    fpart[2 * blockIdx.x] = fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]));
    fpart[2 * blockIdx.x + 1] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
  }
}

// ---------------------------------------------------------------- k_syrk:  partial  sum_f Y_f Y_f^T  and  sum_f Y_f z_f,  Y_f = W_f L_f^-T
// grid (G, npg), block 256.  Per batch of B frames: W rows (12C x 6 per frame) are copied from the linearisation
// records into LDS together with L, 1/diag(L) and z of each frame; every thread then forward-substitutes a few rows
// in place (Y = W L^-T never touches HBM); thread t < 252 owns a 3x4 tile of one 12x12 block pair (ci <= cj):
// 21 block pairs per workgroup, 12 tiles per pair.
__global__ __launch_bounds__(256) void k_syrk(const double* __restrict__ rec, const double* __restrict__ fbuf, const int* __restrict__ pair_ci, const int* __restrict__ pair_cj,
                                              double* __restrict__ spart, double* __restrict__ rpart, int C, int F, int Fpad, int npairs, int fpc, int B) {
  extern __shared__ __align__(16) double s_y[];  // [B][n*6] Y, then [B][34]: L(21) 1/diag(6) z(6) pad
  const int n = 12 * C, n6 = n * 6;
  double* s_f = s_y + (size_t)B * n6;
  const int t = threadIdx.x;
  const int q = blockIdx.y * 21 + t / 12;
  const bool active = (t < 252) && (q < npairs);
  int ra = 0, rb = 0;
  if (active) {
    int tt = t % 12;
    ra = pair_ci[q] * 12 + 3 * (tt / 3);
    rb = pair_cj[q] * 12 + 4 * (tt % 3);
  }
  double acc[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[r][s] = 0.0;
  double racc[2] = {0.0, 0.0};  // rhs rows t, t+256 (blockIdx.y == 0 only)
  const int f0 = blockIdx.x * fpc, f1 = min(F, f0 + fpc);
  for (int fb = f0; fb < f1; fb += B) {
    const int nb = min(B, f1 - fb);
    // gather W of nb consecutive frames out of the wave tiles: for one (camera, element) the nb frames are
    // contiguous doubles, so consecutive threads (b fastest) read contiguous 8*nb-byte runs
    const int nfb = Fpad >> 6;
    for (int i = t; i < nb * C * 72; i += 256) {
      int b = i % nb, ce = i / nb, c = ce / 72, e = ce - c * 72;
      int f = fb + b;
      s_y[(size_t)b * n6 + c * 72 + e] = rec[((size_t)c * nfb + (f >> 6)) * (MCBA_REC * 64) + e * 64 + (f & 63)];
    }
    for (int i = t; i < nb * 27; i += 256) {
      int b = i / 27, k = i - b * 27;
      s_f[b * 34 + (k < 21 ? k : k + 6)] = fbuf[(size_t)(fb + b) * MCBA_FB + k];  // L -> [0,21), z -> [27,33)
    }
    __syncthreads();
    if (t < nb * 6) {
      int b = t / 6, k = t - b * 6;
      s_f[b * 34 + 21 + k] = 1.0 / s_f[b * 34 + k * (k + 1) / 2 + k];
    }
    __syncthreads();
    for (int i = t; i < nb * n; i += 256) {  // Y rows in place: L y = w
      int b = i / n;
      double* w = s_y + (size_t)i * 6;
      const double* Lp = s_f + b * 34;
      double wr[6], yr[6];
#pragma unroll
      for (int k = 0; k < 6; k += 2) { double2 v = *reinterpret_cast<const double2*>(w + k); wr[k] = v.x; wr[k + 1] = v.y; }
      fwd6(Lp, Lp + 21, wr, yr);
#pragma unroll
      for (int k = 0; k < 6; k += 2) *reinterpret_cast<double2*>(w + k) = make_double2(yr[k], yr[k + 1]);
    }
    __syncthreads();
    if (active) {
      for (int b = 0; b < nb; ++b) {
        const double* ya = s_y + (size_t)b * n6 + ra * 6;
        const double* yb = s_y + (size_t)b * n6 + rb * 6;
        double a[3][6], bb[4][6];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int k = 0; k < 6; k += 2) { double2 v = *reinterpret_cast<const double2*>(ya + 6 * r + k); a[r][k] = v.x; a[r][k + 1] = v.y; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int k = 0; k < 6; k += 2) { double2 v = *reinterpret_cast<const double2*>(yb + 6 * s + k); bb[s][k] = v.x; bb[s][k + 1] = v.y; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[r][s] += a[r][k] * bb[s][k];
      }
    }
    if (blockIdx.y == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = t + 256 * j;
        if (row < n) {
          for (int b = 0; b < nb; ++b) {
            const double* yr = s_y + (size_t)b * n6 + row * 6;
            const double* z = s_f + b * 34 + 27;
#pragma unroll
            for (int k = 0; k < 6; ++k) racc[j] += yr[k] * z[k];
          }
        }
      }
    }
    __syncthreads();
  }
  if (active) {
    int tt = t % 12;
    double* o = spart + ((size_t)blockIdx.x * npairs + q) * 144 + (3 * (tt / 3)) * 12 + 4 * (tt % 3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 4; s += 2) *reinterpret_cast<double2*>(o + 12 * r + s) = make_double2(acc[r][s], acc[r][s + 1]);
  }
  if (blockIdx.y == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row = t + 256 * j;
      if (row < n) rpart[(size_t)blockIdx.x * n + row] = racc[j];
    }
  }
}

// ---------------------------------------------------------------- k_reduce_system: fixed-order second stage
// 16 lanes per output double of the reduce buffer (layout in include/mcba.h): lane l sums partials
// l, l+16, l+32, ... (independent loads in flight), then a 4-step xor tree -- a fixed summation order,
// so the result is bit-reproducible run to run (no FP64 atomics anywhere).
__device__ __forceinline__ double strided_sum16(const double* __restrict__ p, size_t stride, int count, int l) {
  double s = 0.0;
  for (int base = 0; base < count; base += 256) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {  // 16 independent loads in flight per lane
      int idx = base + l + 16 * k;
      v[k] = idx < count ? p[(size_t)idx * stride] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}
__device__ __forceinline__ double strided_max16(const double* __restrict__ p, size_t stride, int count, int l) {
  double s = 0.0;
  for (int k = l; k < count; k += 16) s = fmax(s, p[(size_t)k * stride]);
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s = fmax(s, __shfl_xor(s, off, 64));
  return s;
}

__global__ __launch_bounds__(256) void k_reduce_system(const double* __restrict__ gpart, const double* __restrict__ spart, const double* __restrict__ rpart, const double* __restrict__ fpart,
                                                       double* __restrict__ red, int C, int nfb, int G, int npairs, int nfblocks, int rank_slot) {
  const int n = 12 * C;
  const int nsys = n * n + 3 * n + 16;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  if (i >= nsys) return;  // whole 16-lane groups leave together
  double out;
  if (i < n * n) {
    int row = i / n, col = i % n;
    int ci = row / 12, cj = col / 12, li = row % 12, lj = col % 12;
    double s = 0.0;
    if (ci == cj) {
      int a = li <= lj ? li : lj, b = li <= lj ? lj : li;
      s = strided_sum16(gpart + (size_t)ci * nfb * MCBA_GP + tri12(a, b), MCBA_GP, nfb, l);
    }
    int pa = ci <= cj ? ci : cj, pb = ci <= cj ? cj : ci;
    int q = pa * C - (pa * (pa - 1)) / 2 + (pb - pa);
    int loc = ci <= cj ? li * 12 + lj : lj * 12 + li;
    double y = strided_sum16(spart + (size_t)q * 144 + loc, (size_t)npairs * 144, G, l);
    out = s - y;
  } else {
    int j = i - n * n;
    if (j < n) {  // rhs = sum Y z - g_c
      int c = j / 12, lc = j % 12;
      double s = strided_sum16(gpart + (size_t)c * nfb * MCBA_GP + 78 + lc, MCBA_GP, nfb, l);
      double y = strided_sum16(rpart + j, (size_t)n, G, l);
      out = y - s;
    } else if (j < 2 * n) {  // diag U
      int jj = j - n, c = jj / 12, lc = jj % 12;
      out = strided_sum16(gpart + (size_t)c * nfb * MCBA_GP + tri12(lc, lc), MCBA_GP, nfb, l);
    } else if (j < 3 * n) {  // g_c
      int jj = j - 2 * n, c = jj / 12, lc = jj % 12;
      out = strided_sum16(gpart + (size_t)c * nfb * MCBA_GP + 78 + lc, MCBA_GP, nfb, l);
    } else {
      int jj = j - 3 * n;
      out = 0.0;
      if (jj == 0 || jj == 1) out = strided_sum16(gpart + 90 + jj, MCBA_GP, C * nfb, l);  // cost, (camera,frame) pairs with data
      else if (jj == 2) out = strided_sum16(fpart + 1, 2, nfblocks, l);
      else if (jj == 4 + rank_slot) out = strided_max16(fpart, 2, nfblocks, l);
    }
  }
  if (l == 0) red[i] = out;
}

// ---------------------------------------------------------------- k_backsub: frame steps + trial parameters
// lane = frame.  t = g_f + W_f^T d_c with W read from the wave tiles (each load = 64 consecutive frames, 512 B),
// d_f = -(L L^T)^-1 t with the Cholesky factor k_frame_factor left in fbuf, x_dst = x_src + d.
// Per-block partials of  sum d^T(lambda D d - g_f),  sum |d_f|^2,  sum |x_f|^2.
__global__ __launch_bounds__(64) void k_backsub(const double* __restrict__ rec, const double* __restrict__ fbuf, const double* __restrict__ dc, const double* __restrict__ xs,
                                                double* __restrict__ xd, double* __restrict__ bpart, int C, int F, int Fpad, double lambda) {
  const int n = 12 * C, nfb = Fpad >> 6;
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < n; i += 64) xd[i] = xs[i] + dc[i];
  double pred = 0.0, dn2 = 0.0, xn2 = 0.0;
  if (f < F) {
    double t[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < C; ++c) {
      const double* w = rec + ((size_t)c * nfb + blockIdx.x) * (MCBA_REC * 64) + threadIdx.x;
#pragma unroll 4
      for (int lr = 0; lr < 12; ++lr) {
        double d = dc[12 * c + lr];  // wave-uniform: scalar load
        double v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = w[(6 * lr + k) * 64];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = fma(v[k], d, t[k]);
      }
    }
    const double* fbp = fbuf + (size_t)f * MCBA_FB;
    double Lp[21], id[6], gf[6], D[6], y[6], dl[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) Lp[k] = fbp[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { gf[k] = fbp[27 + k]; D[k] = fbp[33 + k]; id[k] = 1.0 / Lp[k * (k + 1) / 2 + k]; t[k] += gf[k]; }
    fwd6(Lp, id, t, y);
    bwd6(Lp, id, y, dl);
    const double* xf = xs + n + 6 * (size_t)f;
    double* xo = xd + n + 6 * (size_t)f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double d = -dl[k], xv = xf[k];
      xo[k] = xv + d;
      pred += d * (lambda * D[k] * d - gf[k]);
      dn2 += d * d;
      xn2 += xv * xv;
    }
  }
  double a = wave_sum63(pred), b = wave_sum63(dn2), cc = wave_sum63(xn2);
  if (threadIdx.x == 63) { bpart[3 * blockIdx.x] = a; bpart[3 * blockIdx.x + 1] = b; bpart[3 * blockIdx.x + 2] = cc; }
}

// trial scalars: [cost, pred_f, dn2_f, xn2_f, n_residuals, 0, 0, 0].  One block of 512 threads:
// wavefront w produces scalar w; its 64 lanes stride over the partials, then a fixed xor tree.
__global__ __launch_bounds__(512) void k_sum_trial(const double* __restrict__ cpart, int ncp, const double* __restrict__ bpart, int nbp, double* __restrict__ out) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const double* p = nullptr;
  int stride = 1, count = 0;
  if (w == 0) { p = cpart; stride = 2; count = ncp; }
  else if (w >= 1 && w <= 3 && bpart) { p = bpart + (w - 1); stride = 3; count = nbp; }
  else if (w == 4) { p = cpart + 1; stride = 2; count = ncp; }
  double s0 = 0.0, s1 = 0.0;
  int k = l;
  for (; k + 64 < count; k += 128) { s0 += p[(size_t)k * stride]; s1 += p[(size_t)(k + 64) * stride]; }
  if (k < count) s0 += p[(size_t)k * stride];
  double s = wave_sum(s0 + s1);
  if (l == 0) out[w] = s;
}

// ---------------------------------------------------------------- k_jacobian: materialised residual Jacobian blocks
// grid (F, C), block 64: one wavefront per (camera, frame), lane = board point (chunks of 64 points).
// Reads the observations in their ORIGINAL (C,F,N,2) layout (kept next to the frame-major copy).
// Output (C,F,N,2,18): per scalar residual [12 camera columns | 6 pose columns] of d(residual)/dx = -d(pred)/dx,
// optionally robust-rescaled.  Rows go through LDS (stride 37 doubles: 2-way bank conflicts at most) so that the
// global stores are contiguous 16 B per lane.
template <int LOSS>
__global__ __launch_bounds__(64) void k_jacobian(const double2* __restrict__ obs_raw, const double* __restrict__ obj, const double* __restrict__ x, double* __restrict__ jac,
                                                 double* __restrict__ res, int C, int F, int N, int Fpad, int robust, double fs2, double ifs2) {
  __shared__ double s_rows[64 * 37];
  const int f = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
  // pose / camera constants: computed by every lane of the wave from the same (uniform) inputs
  CamConst cc;
  make_cam_const(x + 12 * c, cc);
  const double* pose = x + 12 * C + 6 * (size_t)f;
  double pz[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pz[i] = pose[i];
  double Rf[9], Jrf[9];
  rot_and_jr(pz, Rf, Jrf);
  PairConst pc;
  make_pair_const(cc.R, cc.t, Rf, pz + 3, pc);
  ChainConst ch;
  make_chain_const(cc.R, cc.Jr, Rf, Jrf, pz + 3, ch);
  Intr K{cc.fx, cc.fy, cc.cx, cc.cy, cc.k1, cc.k2};
  for (int p0 = 0; p0 < N; p0 += 64) {
    const int p = p0 + lane;
    const int np = min(64, N - p0);
    if (p < N) {
      double2 o2 = obs_raw[((size_t)c * F + f) * N + p];  // raw (C,F,N) layout: lanes = consecutive points, coalesced
      bool vu = is_num(o2.x), vv = is_num(o2.y);
      double Xo[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
      ObsRows o;
      obs_rows(K, pc, Xo, o);
      double ru = o2.x - o.up, rv = o2.y - o.vp;
      double su = -1.0, sv = -1.0;  // residual = obs - pred
      if (robust) {
        double rh, gw, w2;
        loss_weights<LOSS>(ru, fs2, ifs2, rh, gw, w2);
        su = -sqrt(w2);
        loss_weights<LOSS>(rv, fs2, ifs2, rh, gw, w2);
        sv = -sqrt(w2);
      }
      su = vu ? su : 0.0;
      sv = vv ? sv : 0.0;
      double Jcu[12], Jcv[12], Jfu[6], Jfv[6];
      expand_rows(o, ch, Jcu, Jcv, Jfu, Jfv);
      double* sr = s_rows + lane * 37;
#pragma unroll
      for (int k = 0; k < 12; ++k) { sr[k] = su * Jcu[k]; sr[18 + k] = sv * Jcv[k]; }
#pragma unroll
      for (int k = 0; k < 6; ++k) { sr[12 + k] = su * Jfu[k]; sr[30 + k] = sv * Jfv[k]; }
      if (res) *reinterpret_cast<double2*>(res + (((size_t)c * F + f) * N + p) * 2) = make_double2(vu ? ru : 0.0, vv ? rv : 0.0);
    }
    __syncthreads();
    // contiguous tile of np*36 doubles
    double2* out = reinterpret_cast<double2*>(jac + (((size_t)c * F + f) * N + p0) * 36);
    for (int j = lane; j < np * 18; j += 64) {
      int pp = j / 18, kk = (j % 18) * 2;
      out[j] = make_double2(s_rows[pp * 37 + kk], s_rows[pp * 37 + kk + 1]);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- launch wrappers (host)
#define DISPATCH_LOSS(loss, CALL)                                  \
  switch (loss) {                                                  \
    case LOSS_LINEAR: { constexpr int L = LOSS_LINEAR; CALL; } break;   \
    case LOSS_SOFT_L1: { constexpr int L = LOSS_SOFT_L1; CALL; } break; \
    case LOSS_HUBER: { constexpr int L = LOSS_HUBER; CALL; } break;     \
    case LOSS_CAUCHY: { constexpr int L = LOSS_CAUCHY; CALL; } break;   \
    default: { constexpr int L = LOSS_ARCTAN; CALL; } break;            \
  }

void launch_transpose_obs(hipStream_t st, const double* raw, double* obs_t, int C, int F, int N, int Fpad) {
  size_t total = (size_t)C * N * Fpad;
  k_transpose_obs<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(raw), reinterpret_cast<double2*>(obs_t), C, F, N, Fpad);
}

void launch_gram(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, const double* x, double* rec, double* gpart, int C, int N, int Fpad, int split) {
  int nfb = Fpad / 64;
  dim3 block(256);
  double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  if (split) {
    dim3 grid((nfb + 3) / 4, C, 2);
    DISPATCH_LOSS(loss, (k_gram_split<L><<<grid, block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, rec, gpart, C, N, Fpad, nfb, fs2, ifs2)));
  } else {
    dim3 grid((nfb + 3) / 4, C, 1);
    DISPATCH_LOSS(loss, (k_gram<L><<<grid, block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, rec, gpart, C, N, Fpad, nfb, fs2, ifs2)));
  }
}

void launch_cost(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, const double* x, double* cpart, double* res, int C, int F, int N, int Fpad, int nch) {
  int nfb = Fpad / 64;
  dim3 grid((nfb + 3) / 4, C, nch), block(256);
  double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  if (res) {
    DISPATCH_LOSS(loss, (k_cost<L, true><<<grid, block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, cpart, res, C, F, N, Fpad, nfb, nch, fs2, ifs2)));
  } else {
    DISPATCH_LOSS(loss, (k_cost<L, false><<<grid, block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, cpart, res, C, F, N, Fpad, nfb, nch, fs2, ifs2)));
  }
}

void launch_frame_factor(hipStream_t st, const double* rec, double* fbuf, double* fpart, int C, int F, int Fpad, double lambda) {
  k_frame_factor<<<dim3((F + 255) / 256), dim3(256), 0, st>>>(rec, fbuf, fpart, C, F, Fpad, lambda);
}

void launch_syrk(hipStream_t st, const double* rec, const double* fbuf, const int* pair_ci, const int* pair_cj, double* spart, double* rpart, int C, int F, int Fpad, int npairs, int G, int fpc, int B) {
  int npg = (npairs + 20) / 21;
  size_t lds = (size_t)B * (12 * C * 6 + 34) * sizeof(double);
  k_syrk<<<dim3(G, npg), dim3(256), lds, st>>>(rec, fbuf, pair_ci, pair_cj, spart, rpart, C, F, Fpad, npairs, fpc, B);
}

void launch_reduce_system(hipStream_t st, const double* gpart, const double* spart, const double* rpart, const double* fpart, double* red, int C, int nfb, int G, int npairs, int nfblocks, int rank_slot) {
  int n = 12 * C, nsys = n * n + 3 * n + 16;
  k_reduce_system<<<dim3((nsys * 16 + 255) / 256), dim3(256), 0, st>>>(gpart, spart, rpart, fpart, red, C, nfb, G, npairs, nfblocks, rank_slot);
}

void launch_backsub(hipStream_t st, const double* rec, const double* fbuf, const double* dc, const double* xs, double* xd, double* bpart, int C, int F, int Fpad, double lambda) {
  k_backsub<<<dim3(Fpad / 64), dim3(64), 0, st>>>(rec, fbuf, dc, xs, xd, bpart, C, F, Fpad, lambda);
}

void launch_sum_trial(hipStream_t st, const double* cpart, int ncp, const double* bpart, int nbp, double* out) {
  k_sum_trial<<<dim3(1), dim3(512), 0, st>>>(cpart, ncp, bpart, nbp, out);
}

void launch_jacobian(hipStream_t st, int loss, double f_scale, const double* obs_raw, const double* obj, const double* x, double* jac, double* res, int C, int F, int N, int Fpad, int robust) {
  double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  DISPATCH_LOSS(loss, (k_jacobian<L><<<dim3(F, C), dim3(64), 0, st>>>(reinterpret_cast<const double2*>(obs_raw), obj, x, jac, res, C, F, N, Fpad, robust, fs2, ifs2)));
}

int syrk_set_lds_limit(size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace mcba
