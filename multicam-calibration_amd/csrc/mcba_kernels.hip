// mcba_kernels.hip -- gfx950 kernels of the bundle-adjustment hot path (FP64; the per-observation kernels are VALU/HBM
// bound, the Schur SYRK runs on the FP64 matrix cores).
//
// Work decomposition:
//   k_gram / k_cost     one LANE per (camera c, frame f): lanes of a wavefront are 64 consecutive frames of
//                       one camera, the loop runs over the board points.  Observations are stored
//                       [camera][point][frame] so each iteration's load is one coalesced 1 KiB line group;
//                       all accumulation (12x12 local Gram matrix) is lane-local -- no shuffles in the loop.
//                       Camera intrinsics + pose are staged in LDS once per workgroup.  The fused variant's point
//                       loop is branch-free and software-pipelined; rows are accumulated in camera-aligned axes.
//   k_syrk              [the LM accept / reject decision, in every workgroup] + the 6x6 Cholesky factors of the workgroup's
//                       damped frame blocks (z = L^-1 g_f) + per stage of frames: Y = W L^-T built in LDS, S -= Y Y^T on
//                       v_mfma_f64_16x16x4 (the one GEMM-shaped step); the next stage's operands are prefetched meanwhile.
//   k_reduce_system     fixed-order second-stage reduction (deterministic; no FP64 atomics anywhere).
//   k_backsub           64 frames x min(C, 8) wavefronts: frame steps, trial parameters, predicted-reduction terms.
//   k_sum_trial/k_decide  trial sums, accept/reject + damping update + ftol/xtol on the device LM state (mcba_lm.h).
//   (mcba_solve.hip)    k_solve_cam: the reduced camera system factorised and solved by one workgroup.
//   k_jacobian          one wavefront per (camera, frame), one lane per board point; rows transposed through
//                       LDS so the 288 B/observation Jacobian blocks leave as coalesced 16 B/lane stores.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>
#include "mcba_math.h"
#include <algorithm>
#include "mcba_kernels.h"
#include "mcba_lm.h"
#include "mcba_device.h"
#include "mcba_backsub.h"

#ifndef MCBA_GRAM_PIPE
#define MCBA_GRAM_PIPE 1
#endif

namespace mcba {

// Sums over the 64 lanes of K = 3 * 2^m values at once (m <= 5) by recursive halving: at every level a lane hands half of its
// values to its partner and adds the partner's copies of the half it keeps, so level t works on K / 2^(t+1) values instead of
// K -- 3 K (1 - 2^-m) exchanges in all instead of 6 K for K separate butterflies -- and the totals end up SPREAD over the
// lanes: lane l holds the sums of the values  3 (l >> (6 - m)) + {0, 1, 2}  (the lanes that share l >> (6 - m) hold copies).
// Pairings: lanes 32 apart (v_permlane32_swap), rows 16 apart (v_permlane16_swap) -- both exchange in place, no selects --,
// then DPP row_mirror / row_half_mirror / reversed quads / neighbours inside a row.
template <int CTRL>
__device__ __forceinline__ double dpp_perm(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], CTRL, 0xF, 0xF, true);
  b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], CTRL, 0xF, 0xF, true);
  return b.d;
}
template <int LEVEL>
__device__ __forceinline__ double halve_pair(double x, double y, int lane) {  // lanes with the level's bit clear keep x, the others y
  if constexpr (LEVEL <= 1) {
    union { double d; unsigned u[2]; } a, b;
    a.d = x; b.d = y;
    if constexpr (LEVEL == 0) {
      auto r0 = __builtin_amdgcn_permlane32_swap(a.u[0], b.u[0], false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(a.u[1], b.u[1], false, false);
      a.u[0] = r0[0]; b.u[0] = r0[1]; a.u[1] = r1[0]; b.u[1] = r1[1];
    } else {
      auto r0 = __builtin_amdgcn_permlane16_swap(a.u[0], b.u[0], false, false);
      auto r1 = __builtin_amdgcn_permlane16_swap(a.u[1], b.u[1], false, false);
      a.u[0] = r0[0]; b.u[0] = r0[1]; a.u[1] = r1[0]; b.u[1] = r1[1];
    }
    return a.d + b.d;
  } else {
    constexpr int BIT = LEVEL == 2 ? 8 : LEVEL == 3 ? 4 : 2;
    constexpr int CTRL = LEVEL == 2 ? 0x140 : LEVEL == 3 ? 0x141 : 0x1B;  // row_mirror, row_half_mirror, quad_perm [3,2,1,0]
    const bool up = (lane & BIT) != 0;
    const double keep = up ? y : x, send = up ? x : y;
    return keep + dpp_perm<CTRL>(send);
  }
}
template <int LEVEL>
__device__ __forceinline__ double pair_sum(double v) {  // plain butterfly of one level (the levels left over when K < 96)
  if constexpr (LEVEL == 4) return v + dpp_perm<0x1B>(v);
  else return v + dpp_perm<0xB1>(v);  // level 5: neighbours
}
template <int K, int LEVEL = 0>
__device__ __forceinline__ void wave_reduce_scatter(double* v, double (&out)[3], int lane) {
  if constexpr (K > 3) {
    constexpr int H = K / 2;
#pragma unroll
    for (int k = 0; k < H; ++k) v[k] = halve_pair<LEVEL>(v[k], v[k + H], lane);
    wave_reduce_scatter<H, LEVEL + 1>(v, out, lane);
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double t = v[r];
      if constexpr (LEVEL <= 4) t = pair_sum<4>(t);
      t = pair_sum<5>(t);
      out[r] = t;
    }
    static_assert(LEVEL == 4 || LEVEL == 5, "K must be 48 or 96");
  }
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
template <int LOSS, bool UNIT = false>
__device__ __forceinline__ void obs_weights(double r, bool valid, double fs2, double ifs2, double cfl, double& cost, double& w2, double& g) {
  double rh, gw, ww;
  loss_weights<LOSS, UNIT>(r, fs2, ifs2, rh, gw, ww);
  cost += valid ? rh : 0.0;
  w2 = valid ? lm_weight(gw, ww, cfl) : 0.0;
  g = valid ? gw * r : 0.0;
}

// LOSS_TABLE: the three numbers come from the caller's table (mcba_set_loss_table: 0.5 f_scale^2 rho, rho', scipy's J_scale^2 -- evaluated
// by the caller's function at the residuals of the point being linearised), not from r
__device__ __forceinline__ void obs_weights_table(double r, bool valid, double rh, double gw, double ww, double cfl, double& cost, double& w2, double& g) {
  cost += valid ? rh : 0.0;
  w2 = valid ? lm_weight(gw, ww, cfl) : 0.0;
  g = valid ? gw * r : 0.0;
}

// ---------------------------------------------------------------- observation re-layout
// raw (C,F,N,2) -> obs_t [C][N][Fpad] (u,v); frames f >= F are NaN (= missing, contribute nothing).
// Through an LDS tile of 64 frames x 16 points: reads are 256 contiguous bytes per frame (16 points), writes 1 KB per point (64
// frames).  (One thread per output element, as it was, read 16 bytes at a stride of 16 N: 326 MB fetched for the 52 MB array at
// 6 x 10 000 x 54 by FETCH_SIZE, 2.16 GB for 482 MB at 24 x 6 250 x 200.)
__global__ __launch_bounds__(256) void k_transpose_obs(const double2* __restrict__ raw, double2* __restrict__ obs_t, int C, int F, int N, int Fpad) {
  __shared__ double2 tile[64][17];
  const int fb = blockIdx.x, c = blockIdx.y, p0 = 16 * blockIdx.z;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double2 nanv;
  nanv.x = nanv.y = __builtin_nan("");
#pragma unroll
  for (int fr = 0; fr < 4; ++fr) {
    const int f = fb * 64 + fr * 16 + ty, p = p0 + tx;
    tile[fr * 16 + ty][tx] = (f < F && p < N) ? raw[((size_t)c * F + f) * N + p] : nanv;
  }
  __syncthreads();
  const int fl = threadIdx.x & 63, pp = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = p0 + pp + 4 * k;
    if (p < N) obs_t[((size_t)c * N + p) * Fpad + fb * 64 + fl] = tile[fl][pp + 4 * k];
  }
}

// ---------------------------------------------------------------- k_gram: linearise
// grid (ceil(nfb/4), C, 2), block 256 = 4 wavefronts = 4 frame blocks of one camera; blockIdx.z is the ROLE
// (mcba_math.h: role A = [A|P] block -> V, g_f, W rows of rho/t, U_(rho,t)x(rho,t); role B = intrinsics blocks).
// Splitting the 87 accumulators over two wavefronts keeps each under 256 VGPRs, so two waves share a SIMD and
// hide each other's FP64 / memory latency; the roles never exchange data.
// FAST (fused variant only): planar board (every z = 0) and f_scale = 1 -- the reference's own set-up -- known at compile time:
// seven FP64 instructions per point-observation less (products with 0.0 / 1.0 dropped; results equal to round-off).
// MODE (fused variant only; round 3): 0 the whole linearisation of one (camera, frame block); 1 POINT CHUNK -- the points [p_lo, p_hi)
// only, and instead of the expansion the 88 raw per-lane sums (local Gram accumulators, cost, data flag) go to `chunk` --; 2 COMBINE --
// no point loop: the sums of `nchunk` chunks are added up in chunk order and expanded.  1 + 2 replace the split-role tail launch of
// shards that are not a whole number of rounds of the 1 024 wavefront slots: the remainder's POINTS are spread over the idle SIMDs.
// MODE 3 (round 4): POINT SPLIT INSIDE THE WORKGROUP -- the NPW wavefronts of one (camera, frame block) each run the fused loop over
// 1 / NPW of the board's points, then meet in LDS: the role-A sums (ee, he, cost, data flag: 29 per lane) of the others go to the
// group's wavefront 0, the role-B sums (ii, ie, hi: 59) to its wavefront 1, each adds them IN PART ORDER and runs its role's half of
// the expansion (the two halves are independent, see mcba_math.h) -- no second launch and no trip of the raw sums through HBM.  For
// every shape that is not a whole number of rounds of the 1 024 wavefront slots: few (camera, frame block) items, or a short last round.
constexpr int kGramRaw = 88;  // ee 21 | he 6 | cost | ii 17 | ie 36 | hi 6 | any
constexpr int kGramRawA = 29, kGramRawB = 59;  // what role A / role B of the expansion needs of them
// doubles of the LDS exchange area of a point-split workgroup (gram_body MODE 3): npw wavefronts per group, with / without role B
__host__ __device__ constexpr int gram_xch_doubles(int npw, bool with_b) {
  return npw == 4 ? (with_b ? (4 * kGramRawA + 3 * 32 + 3 * 27) * 64 : 4 * kGramRawA * 64) : (4 / npw) * (npw - 1) * (with_b ? kGramRaw : kGramRawA) * 64;
}
template <int LOSS, int ROLE, bool FAST = false, int MODE = 0, int NPW = 1>
__device__ __forceinline__ void gram_body(const CamConst& s_cam, const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x,
                                          double* __restrict__ rec, double* __restrict__ gpart, int c, int fb, int lane, int C, int N, int Fpad, int nfb, double fs2, double ifs2, double cfl,
                                          const double (&pz0)[6], const double2 (&pre)[4], double* s_cost, int nrun, int p_lo = 0, int p_hi = -1, double2* chunk = nullptr, int nchunk = 1,
                                          size_t chunk_stride = 0, const double2* __restrict__ ltab = nullptr) {
  static_assert(LOSS != LOSS_TABLE || (ROLE != 2 && MODE == 0), "the tabulated loss runs the split-role kernel (k_gram_table)");
  if (p_hi < 0) p_hi = N;
  static_assert(MODE == 0 || ((ROLE == 2 || (MODE == 3 && ROLE == 0)) && MCBA_GRAM_PIPE), "point chunks exist for the pipelined loop only (both roles; role A alone in MODE 3: intrinsics held fixed)");
  static_assert(MODE == 3 ? (NPW == 2 || NPW == 4) : NPW == 1, "NPW wavefronts share a (camera, frame block) in MODE 3 only");
  // pz0: this lane's frame pose, pre: its first four observations -- loaded by the kernel before the camera constants were
  // staged (those loads, the LM state and the camera rows are all in flight together: one memory round trip at the start)
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
  double abc[3];  // sin t / t, (1 - cos t) / t^2, (t - sin t) / t^3 of the frame's rotation: needed again after the point loop
  {
    double Rf[9];
    rot_only(pz0, Rf, abc);
    make_pair_const(Rc, tc, Rf, pz0 + 3, pc);
  }

  constexpr bool DO_A = ROLE != 1, DO_B = ROLE != 0;
  GramA ga;
  GramB gb;
  if (DO_A) gram_zero(ga);
  if (DO_B) gram_zero(gb);
  double cost = 0.0;
  bool any = false;
  // Observation loads run PF points ahead of their use; the ring is indexed statically by unrolling the loop PF times.
  // (Measured: depth 1, 2 and 4 perform the same -- the loop is FP64-issue bound, not latency bound.)
  constexpr int PF = 2;
  const double2* op = obs_t + (size_t)c * N * Fpad + f;
  double2 ring[PF];
  double xring[PF][3];  // board points travel with the ring: wave-uniform scalar loads issued PF points ahead as well
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int pj = min(j, N - 1);
    ring[j] = pre[j];
    xring[j][0] = obj[3 * pj]; xring[j][1] = obj[3 * pj + 1]; xring[j][2] = obj[3 * pj + 2];
  }
  auto point = [&](double2 o2, const double Xo[3], int pi) {
    bool vu = is_num(o2.x), vv = is_num(o2.y);
    if (vu || vv) {
      any = true;
      ObsCommon q;
      obs_common(K, pc, Xo, q);
      double wu2, wv2, gu, gv;
      if constexpr (LOSS == LOSS_TABLE) {
        // three planes [C][N][Fpad] of (u, v) pairs, laid out as the observations are
        const size_t plane = (size_t)C * N * Fpad, at = (size_t)c * N * Fpad + (size_t)pi * Fpad + f;
        const double2 t0 = ltab[at], t1 = ltab[plane + at], t2 = ltab[2 * plane + at];
        obs_weights_table(o2.x - q.up, vu, t0.x, t1.x, t2.x, cfl, cost, wu2, gu);
        obs_weights_table(o2.y - q.vp, vv, t0.y, t1.y, t2.y, cfl, cost, wv2, gv);
      } else {
      obs_weights<LOSS>(o2.x - q.up, vu, fs2, ifs2, cfl, cost, wu2, gu);
      obs_weights<LOSS>(o2.y - q.vp, vv, fs2, ifs2, cfl, cost, wv2, gv);
      }
      {  // u row, completely, before the v row exists: one [A|P] row live at a time
        double E[6];
        obs_row_cam<0>(q, E);
        if (DO_A) gram_add_row<0>(ga, E, wu2, gu);
        if (DO_B) { double l4 = q.fa * q.s; gram_add_row<0>(gb, E, wu2, gu, q.a * q.d, l4, l4 * q.s); }
      }
      {
        double E[6];
        obs_row_cam<1>(q, E);
        if (DO_A) gram_add_row<1>(ga, E, wv2, gv);
        if (DO_B) { double l4 = q.fb * q.s; gram_add_row<1>(gb, E, wv2, gv, q.b * q.d, l4, l4 * q.s); }
      }
    }
  };
  if constexpr ((ROLE == 2 || MODE == 3) && MCBA_GRAM_PIPE) {
    // Software-pipelined, branch-free point loop (fused variant, one wavefront per SIMD): the projection of point p + 1 --
    // a serial chain (rotate, reciprocal, distortion polynomial) -- is issued next to the 180 independent accumulator
    // updates of point p, which is what fills the FP64 pipe when no second wavefront is there to do it.  Lanes without
    // an observation project a harmless point and add exact zeros.
    constexpr int RD = 4;  // observation ring: points p .. p + RD - 1 resident
    double2 r4[RD];
    double x4[RD][3];
#pragma unroll
    for (int j = 0; j < RD; ++j) {
      int pj = min(p_lo + j, p_hi - 1);
      if constexpr (MODE == 3) pj = max(pj, 0);  // (a piece of a board with fewer points than wavefronts may be empty)
      r4[j] = pre[j];
      x4[j][0] = obj[3 * pj]; x4[j][1] = obj[3 * pj + 1]; x4[j][2] = obj[3 * pj + 2];
    }
    ObsLead qc;
    obs_lead<true, FAST>(pc, x4[0], qc, is_num(r4[0].x) || is_num(r4[0].y));
    auto accumulate = [&](double2 o2, const double Xo[3], const ObsLead& ql) {
      ObsCommon q;
      obs_finish(K, ql, q);
      const bool vu = is_num(o2.x), vv = is_num(o2.y);
      any = any || vu || vv;
      double wu2, wv2, gu, gv;
      obs_weights<LOSS, FAST>(o2.x - q.up, vu, fs2, ifs2, cfl, cost, wu2, gu);
      obs_weights<LOSS, FAST>(o2.y - q.vp, vv, fs2, ifs2, cfl, cost, wv2, gv);
      {
        double E[6];
        obs_row_cam<0>(q, E);
        gram_add_row<0>(ga, E, wu2, gu);
        if constexpr (DO_B) {
        double l4 = q.fa * q.s;
        gram_add_row<0>(gb, E, wu2, gu, q.a * q.d, l4, l4 * q.s);
        }
      }
      {
        double E[6];
        obs_row_cam<1>(q, E);
        gram_add_row<1>(ga, E, wv2, gv);
        if constexpr (DO_B) {
        double l4 = q.fb * q.s;
        gram_add_row<1>(gb, E, wv2, gv, q.b * q.d, l4, l4 * q.s);
        }
      }
    };
    int p = p_lo;
    if constexpr (MODE != 2)
    for (; p + RD <= p_hi; p += RD) {
#pragma unroll
      for (int j = 0; j < RD; ++j) {
        const int jn = (j + 1) % RD;
        ObsLead qn;
        obs_lead<true, FAST>(pc, x4[jn], qn, is_num(r4[jn].x) || is_num(r4[jn].y));  // point p + j + 1 (clamped duplicate at the very end)
        const double2 o2 = r4[j];
        const double Xo[3] = {x4[j][0], x4[j][1], x4[j][2]};
        const int pn = min(p + j + RD, p_hi - 1);
        r4[j] = op[(size_t)pn * Fpad];
        x4[j][0] = obj[3 * pn]; x4[j][1] = obj[3 * pn + 1]; x4[j][2] = obj[3 * pn + 2];
        accumulate(o2, Xo, qc);
        qc = qn;
      }
    }
    if constexpr (MODE != 2)
#pragma unroll
    for (int j = 0; j < RD; ++j) {  // remainder (N not a multiple of 4): same rotation of the ring, no refill
      if (p + j < p_hi) {
        const int jn = (j + 1) % RD;
        ObsLead qn;
        obs_lead<true, FAST>(pc, x4[jn], qn, is_num(r4[jn].x) || is_num(r4[jn].y));
        accumulate(r4[j], x4[j], qc);
        qc = qn;
      }
    }
  } else {
  int p = 0;
  for (; p + PF <= N; p += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      double2 o2 = ring[j];
      double Xo[3] = {xring[j][0], xring[j][1], xring[j][2]};
      const int pn = min(p + j + PF, N - 1);
      ring[j] = op[(size_t)pn * Fpad];
      xring[j][0] = obj[3 * pn]; xring[j][1] = obj[3 * pn + 1]; xring[j][2] = obj[3 * pn + 2];
      point(o2, Xo, p + j);
    }
  }
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (p + j < N) point(ring[j], xring[j], p + j);
  }

  if constexpr (MODE == 1) {  // point chunk: the raw sums, [k / 2][lane] double2 rows (1 KiB each), nothing else
    double raw[kGramRaw];
#pragma unroll
    for (int i = 0; i < 21; ++i) raw[i] = ga.ee[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { raw[21 + i] = ga.he[i]; raw[81 + i] = gb.hi[i]; }
    raw[27] = cost;
#pragma unroll
    for (int i = 0; i < 17; ++i) raw[28 + i] = gb.ii[i];
#pragma unroll
    for (int i = 0; i < 36; ++i) raw[45 + i] = gb.ie[i];
    raw[87] = any ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < kGramRaw; k += 2) chunk[(k >> 1) * 64] = make_double2(raw[k], raw[k + 1]);
    return;
  }
  if constexpr (MODE == 2) {
    // combine: chunk sums in chunk order, THREE chunks (132 double2 rows of 1 KiB) in flight per round trip -- a dependent global
    // round trip costs ~2 us here, the data next to nothing
    constexpr int HR = kGramRaw / 2, CB = 3;
    double acc[kGramRaw];
#pragma unroll
    for (int k = 0; k < kGramRaw; ++k) acc[k] = 0.0;
    for (int c0 = 0; c0 < nchunk; c0 += CB) {
      double2 v[CB][HR];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const double2* src = chunk + (size_t)min(c0 + j, nchunk - 1) * chunk_stride;  // (clamped duplicates are not added)
#pragma unroll
        for (int k = 0; k < HR; ++k) v[j][k] = src[k * 64];
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < nchunk) {
#pragma unroll
          for (int k = 0; k < HR; ++k) { acc[2 * k] += v[j][k].x; acc[2 * k + 1] += v[j][k].y; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 21; ++i) ga.ee[i] = acc[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { ga.he[i] = acc[21 + i]; gb.hi[i] = acc[81 + i]; }
    cost = acc[27];
#pragma unroll
    for (int i = 0; i < 17; ++i) gb.ii[i] = acc[28 + i];
#pragma unroll
    for (int i = 0; i < 36; ++i) gb.ie[i] = acc[45 + i];
    any = acc[87] != 0.0;
  }
  if constexpr (MODE == 3) {
    // ---- the wavefronts of the group meet in LDS (see the head of this function): xs = [group][A: NPW - 1 sources][29][64] then
    // [group][B: NPW - 1 sources][59][64] doubles; every access is 64 consecutive doubles (conflict-free ds_*_b64)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), part = wv % NPW, grp = wv / NPW;
    if constexpr (NPW == 4) {
      // FOUR finishing roles, one per wavefront, so that the expansion and the cross-lane reduction of the 92 per-wavefront sums --
      // about a third of this kernel's time on a 54-point board when one or two wavefronts do it -- are spread over the four SIMDs:
      //   part 0: role A -> U rows 6..11, g_c[6..12), cost, pairs with data (29 sums, reduced over the lanes)
      //   part 3: role A -> the record's W rows 6..11, V, g_f (no reduction)
      //   part 1: role B, intrinsics rows 0..2 -> U rows 0..2, g_c[0..3) (36 sums), W rows 0..2
      //   part 2: role B, intrinsics rows 3..5 -> U rows 3..5, g_c[3..6) (27 sums), W rows 3..5
      // LDS: A sums [4 parts][29][64] (both readers add all four in part order), B-low (ii 0..10, ie rows 0..2, hi 0..2: 32) from
      // parts 0, 2, 3 and B-high (ii 11..16, ie rows 3..5, hi 3..5: 27) from parts 0, 1, 3 -- 150 016 bytes.
      constexpr int NBL = 32, NBH = 27;
      double* xA = reinterpret_cast<double*>(chunk) + lane;
      double* xL = xA + 4 * kGramRawA * 64;
      double* xH = xL + 3 * NBL * 64;
      {
        double* d = xA + part * (kGramRawA * 64);
#pragma unroll
        for (int i = 0; i < 21; ++i) d[i * 64] = ga.ee[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) d[(21 + i) * 64] = ga.he[i];
        d[27 * 64] = cost;
        d[28 * 64] = any ? 1.0 : 0.0;
      }
      if (DO_B && part != 1) {
        double* d = xL + (part == 0 ? 0 : part - 1) * (NBL * 64);
#pragma unroll
        for (int i = 0; i < 11; ++i) d[i * 64] = gb.ii[i];
#pragma unroll
        for (int i = 0; i < 18; ++i) d[(11 + i) * 64] = gb.ie[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[(29 + i) * 64] = gb.hi[i];
      }
      if (DO_B && part != 2) {
        double* d = xH + (part == 3 ? 2 : part) * (NBH * 64);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i * 64] = gb.ii[11 + i];
#pragma unroll
        for (int i = 0; i < 18; ++i) d[(6 + i) * 64] = gb.ie[18 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[(24 + i) * 64] = gb.hi[3 + i];
      }
      double pz[6];  // the frame's pose again (not kept over the point loop): requested before the barrier, needed after it
#pragma unroll
      for (int i = 0; i < 6; ++i) pz[i] = pose[i];
      __syncthreads();
      if (!DO_B && (part == 1 || part == 2)) return;  // intrinsics held fixed: role A alone (no timing stamps from these two)
      ChainConst ch;
      {
        double Rf[9], Jrf[9], Jrc[9];
        rot_and_jr(pz, Rf, Jrf, abc);
#pragma unroll
        for (int i = 0; i < 9; ++i) Jrc[i] = uni(s_cam.Jr[i]);
        make_chain_const(Rc, Jrc, Rf, Jrf, pz + 3, ch);
        chain_to_cam_rows(pc.Rcf, ch);
      }
      double2* r2 = reinterpret_cast<double2*>(rec + ((size_t)c * nfb + fb) * (MCBA_REC * 64)) + lane;
      double red[48];
#pragma unroll
      for (int i = 0; i < 48; ++i) red[i] = 0.0;
      int kofs_lo = 0, kofs_hi = 0, nlo = 0, nown = 0;  // compact index j of this part's sums -> row k of gpart: j < nlo ? kofs_lo + j : kofs_hi + j
      if (part == 0 || part == 3) {
        GramA gs;
#pragma unroll
        for (int i = 0; i < 21; ++i) gs.ee[i] = xA[i * 64];
#pragma unroll
        for (int i = 0; i < 6; ++i) gs.he[i] = xA[(21 + i) * 64];
        double cs = xA[27 * 64];
        bool an = xA[28 * 64] != 0.0;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          const double* sA = xA + j * (kGramRawA * 64);
#pragma unroll
          for (int i = 0; i < 21; ++i) gs.ee[i] += sA[i * 64];
#pragma unroll
          for (int i = 0; i < 6; ++i) gs.he[i] += sA[(21 + i) * 64];
          cs += sA[27 * 64];
          an = an || sA[28 * 64] != 0.0;
        }
        double U[78], gc[12], W[72], V[21], gf[6];
        gram_expand(gs, ch, U, gc, W, V, gf);
        if (part == 3) {
#pragma unroll
          for (int i = 36; i < 72; i += 2) r2[(i >> 1) * 64] = make_double2(W[i], W[i + 1]);
#pragma unroll
          for (int i = 0; i < 20; i += 2) r2[(36 + (i >> 1)) * 64] = make_double2(V[i], V[i + 1]);
          r2[46 * 64] = make_double2(V[20], gf[0]);
          r2[47 * 64] = make_double2(gf[1], gf[2]);
          r2[48 * 64] = make_double2(gf[3], gf[4]);
          r2[49 * 64] = make_double2(gf[5], 0.0);
        } else {
#pragma unroll
          for (int a = 6; a < 12; ++a) {
#pragma unroll
            for (int b = a; b < 12; ++b) red[tri12(a, b) - 57] = U[tri12(a, b)];
            red[78 + a - 63] = gc[a];
          }
          red[27] = cs;
          red[28] = an ? 1.0 : 0.0;
          kofs_lo = 57; kofs_hi = 63; nlo = 21; nown = 29;
        }
      } else if constexpr (DO_B) {
        GramB gs;
        gram_zero(gs);
        if (part == 1) {  // ((part 0 + own) + part 2) + part 3
#pragma unroll
          for (int i = 0; i < 11; ++i) gs.ii[i] = xL[i * 64] + gb.ii[i];
#pragma unroll
          for (int i = 0; i < 18; ++i) gs.ie[i] = xL[(11 + i) * 64] + gb.ie[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) gs.hi[i] = xL[(29 + i) * 64] + gb.hi[i];
#pragma unroll
          for (int j = 1; j < 3; ++j) {
            const double* sB = xL + j * (NBL * 64);
#pragma unroll
            for (int i = 0; i < 11; ++i) gs.ii[i] += sB[i * 64];
#pragma unroll
            for (int i = 0; i < 18; ++i) gs.ie[i] += sB[(11 + i) * 64];
#pragma unroll
            for (int i = 0; i < 3; ++i) gs.hi[i] += sB[(29 + i) * 64];
          }
        } else {  // ((part 0 + part 1) + own) + part 3
#pragma unroll
          for (int i = 0; i < 6; ++i) gs.ii[11 + i] = ((xH[i * 64] + xH[(NBH + i) * 64]) + gb.ii[11 + i]) + xH[(2 * NBH + i) * 64];
#pragma unroll
          for (int i = 0; i < 18; ++i) gs.ie[18 + i] = ((xH[(6 + i) * 64] + xH[(NBH + 6 + i) * 64]) + gb.ie[18 + i]) + xH[(2 * NBH + 6 + i) * 64];
#pragma unroll
          for (int i = 0; i < 3; ++i) gs.hi[3 + i] = ((xH[(24 + i) * 64] + xH[(NBH + 24 + i) * 64]) + gb.hi[3 + i]) + xH[(2 * NBH + 24 + i) * 64];
        }
        double U[78], gc[12], W[72];
        gram_expand(gs, ch, U, gc, W);
        if (part == 1) {
#pragma unroll
          for (int i = 0; i < 18; i += 2) r2[(i >> 1) * 64] = make_double2(W[i], W[i + 1]);
#pragma unroll
          for (int j = 0; j < 33; ++j) red[j] = U[j];  // rows 0..2 of the upper triangle are its first 33 entries
#pragma unroll
          for (int a = 0; a < 3; ++a) red[33 + a] = gc[a];
          kofs_lo = 0; kofs_hi = 78 - 33; nlo = 33; nown = 36;
        } else {
#pragma unroll
          for (int i = 18; i < 36; i += 2) r2[(i >> 1) * 64] = make_double2(W[i], W[i + 1]);
#pragma unroll
          for (int j = 0; j < 24; ++j) red[j] = U[33 + j];  // rows 3..5: entries 33..56
#pragma unroll
          for (int a = 0; a < 3; ++a) red[24 + a] = gc[3 + a];
          kofs_lo = 33; kofs_hi = 81 - 24; nlo = 24; nown = 27;
        }
      }
      if (part != 3) {
        double o3[3];
        wave_reduce_scatter<48>(red, o3, lane);
        if ((lane & 3) == 0) {  // lanes that share lane >> 2 hold copies: the first of them stores
          double* gp = gpart + (size_t)c * MCBA_GP * nfb + fb;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int j = 3 * (lane >> 2) + r;
            if (j < nown) gp[(size_t)(j < nlo ? kofs_lo + j : kofs_hi + j) * nfb] = o3[r];
          }
        }
      }
    } else {
    double* xa = reinterpret_cast<double*>(chunk) + (size_t)grp * ((NPW - 1) * (DO_B ? kGramRaw : kGramRawA) * 64) + lane;
    double* xb = xa + (NPW - 1) * kGramRawA * 64;
    if (part != 0) {  // role-A sums -> the group's wavefront 0 (source slot part - 1)
      double* d = xa + (part - 1) * (kGramRawA * 64);
#pragma unroll
      for (int i = 0; i < 21; ++i) d[i * 64] = ga.ee[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[(21 + i) * 64] = ga.he[i];
      d[27 * 64] = cost;
      d[28 * 64] = any ? 1.0 : 0.0;
    }
    if (DO_B && part != 1) {  // role-B sums -> the group's wavefront 1 (source slots: part 0 -> 0, part j >= 2 -> j - 1)
      double* d = xb + (part == 0 ? 0 : part - 1) * (kGramRawB * 64);
#pragma unroll
      for (int i = 0; i < 17; ++i) d[i * 64] = gb.ii[i];
#pragma unroll
      for (int i = 0; i < 36; ++i) d[(17 + i) * 64] = gb.ie[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[(53 + i) * 64] = gb.hi[i];
    }
    __syncthreads();  // (wavefronts past the last frame block have ended: the barrier does not wait for them)
    if (part == 0) {  // parts 0, 1, 2, ... in order
#pragma unroll
      for (int j = 1; j < NPW; ++j) {
        const double* sA = xa + (j - 1) * (kGramRawA * 64);
#pragma unroll
        for (int i = 0; i < 21; ++i) ga.ee[i] += sA[i * 64];
#pragma unroll
        for (int i = 0; i < 6; ++i) ga.he[i] += sA[(21 + i) * 64];
        cost += sA[27 * 64];
        any = any || sA[28 * 64] != 0.0;
      }
      {
#define GF_ROLE 0
#define GF_A true
#define GF_B false
#define GF_PRESUM false
#include "mcba_gram_finish.inc"
#undef GF_ROLE
#undef GF_A
#undef GF_B
#undef GF_PRESUM
      }
    } else if (DO_B && part == 1) {
      {  // part 0 + own, then parts 2, 3: the same order of additions as above
        const double* sB = xb;
#pragma unroll
        for (int i = 0; i < 17; ++i) gb.ii[i] = sB[i * 64] + gb.ii[i];
#pragma unroll
        for (int i = 0; i < 36; ++i) gb.ie[i] = sB[(17 + i) * 64] + gb.ie[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) gb.hi[i] = sB[(53 + i) * 64] + gb.hi[i];
      }
#pragma unroll
      for (int j = 2; j < NPW; ++j) {
        const double* sB = xb + (j - 1) * (kGramRawB * 64);
#pragma unroll
        for (int i = 0; i < 17; ++i) gb.ii[i] += sB[i * 64];
#pragma unroll
        for (int i = 0; i < 36; ++i) gb.ie[i] += sB[(17 + i) * 64];
#pragma unroll
        for (int i = 0; i < 6; ++i) gb.hi[i] += sB[(53 + i) * 64];
      }
      {
#define GF_ROLE 1
#define GF_A false
#define GF_B true
#define GF_PRESUM false
#include "mcba_gram_finish.inc"
#undef GF_ROLE
#undef GF_A
#undef GF_B
#undef GF_PRESUM
      }
    }
    }
    return;
  }
#define GF_ROLE ROLE
#define GF_A DO_A
#define GF_B DO_B
#define GF_PRESUM true
#include "mcba_gram_finish.inc"
#undef GF_ROLE
#undef GF_A
#undef GF_B
#undef GF_PRESUM
}

// Common start of the two k_gram kernels.  Which parameter slot / record buffer is the current one lives in the device LM
// state, the camera row and the frame poses live in that slot: read one after the other that is three dependent memory round
// trips (~2 us each, the data was just written by kernels on other XCDs) before the first FMA.  Instead the state, the
// camera row and this lane's pose of BOTH slots and the lane's first four observations (slot-independent) are requested
// together, and the right copies are picked once the state has arrived.
struct GramStart {
  const double* x;
  double* rec;
  double* gpart;
  double pz[6];
  double2 pre[4];
  int fb, lane;
  bool run;
  double cfl;  // curvature floor of this linearisation: the LM state's (device-resident loops) or the launch's (Sel.cfl)
};
__device__ __forceinline__ void gram_start(GramStart& g, CamConst& s_cam, const double2* __restrict__ obs_t, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1, double* rec0, double* rec1,
                                           double* gp0, double* gp1, int C, int N, int Fpad, int fb0, int fb1, int p_lo = 0, int p_hi = -1, int npw = 1,
                                           int bxo = -1, int co = -1) {  // bxo, co: workgroup coordinates decoded by the caller (k_gram_mixed) instead of blockIdx.x / .y
  if (p_hi < 0) p_hi = N;
  const int c = co >= 0 ? co : (int)blockIdx.y;
  const int bx = bxo >= 0 ? bxo : (int)blockIdx.x;
  const int wave = threadIdx.x >> 6;
  g.lane = threadIdx.x & 63;
  g.fb = fb0 + bx * ((blockDim.x >> 6) / npw) + wave / npw;  // this launch covers the frame blocks [fb0, fb1), one per wavefront (point split: per npw wavefronts)
  const bool have = g.fb < fb1;
  const int fbc = have ? g.fb : fb1 - 1;
  double st3 = 0.0, st14 = 0.0, st15 = 0.0, st25 = sl.cfl;
  if (sl.lms) { st3 = sl.lms[3]; st14 = sl.lms[MCBA_LM_SKIP]; st15 = sl.lms[MCBA_LM_DONE]; st25 = sl.lms[MCBA_LM_CFL]; }
  const size_t po = (size_t)12 * C + 6 * ((size_t)fbc * 64 + g.lane);
  double pa[6], pb[6], ca[12], cb[12];
#pragma unroll
  for (int i = 0; i < 6; ++i) { pa[i] = x0[po + i]; pb[i] = x1[po + i]; }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) { ca[i] = x0[12 * c + i]; cb[i] = x1[12 * c + i]; }
  }
  const double2* op = obs_t + (size_t)c * N * Fpad + (size_t)fbc * 64 + g.lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) g.pre[j] = op[(size_t)min(p_lo + j, p_hi - 1) * Fpad];
  // ---- the state has arrived: sel_active(sl, true) / sel_index(sl)
  const bool active = !sl.lms || (st15 == 0.0 && st14 == 0.0);
  const bool spec = sl.spec && st14 == 0.0;
  const int sidx = sl.lms ? ((static_cast<int>(st3) ^ sl.idx ^ (spec ? 1 : 0)) & 1) : sl.idx;
  g.x = sidx ? x1 : x0;
  g.rec = sidx ? rec1 : rec0;
  g.gpart = sidx ? gp1 : gp0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g.pz[i] = sidx ? pb[i] : pa[i];
  if (active && threadIdx.x == 0) {
    double cm[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) cm[i] = sidx ? cb[i] : ca[i];
    make_cam_const(cm, s_cam);
  }
  g.cfl = st25;
  g.run = active;
  if (active) __syncthreads();  // (uniform: the state is the same for every thread)
  g.run = active && have;
}

// Split roles: grid.z = 2, <= 256 VGPRs, two waves per SIMD.
template <int LOSS>
__global__ __launch_bounds__(256, 2) void k_gram_split(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                       double* __restrict__ rec0, double* __restrict__ rec1, double* __restrict__ gp0, double* __restrict__ gp1, int C, int N, int Fpad, int nfb, int fb0, int fb1,
                                                       double fs2, double ifs2) {
  __shared__ CamConst s_cam;
  __shared__ double s_cost[8];
  GramStart g;
  gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, fb0, fb1);
  if (!g.run) return;
  const int c = blockIdx.y;
  const int nrun = min(4, fb1 - (fb0 + (int)blockIdx.x * 4));  // wavefronts of this workgroup that have a frame block
  if (blockIdx.z == 0) gram_body<LOSS, 0>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, c, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, nrun);
  else gram_body<LOSS, 1>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, c, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, nrun);
}

// The caller's loss, tabulated (LOSS_TABLE; least_squares' callable `loss`): the split-role kernel with the table behind one more argument.
// An off-default path driven from the host (solver.py evaluates the caller's function between the launches): the plain point loop.
__global__ __launch_bounds__(256, 2) void k_gram_table(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                       double* __restrict__ rec0, double* __restrict__ rec1, double* __restrict__ gp0, double* __restrict__ gp1, int C, int N, int Fpad, int nfb, int fb0, int fb1,
                                                       const double2* __restrict__ ltab) {
  __shared__ CamConst s_cam;
  __shared__ double s_cost[8];
  GramStart g;
  gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, fb0, fb1);
  if (!g.run) return;
  const int c = blockIdx.y;
  const int nrun = min(4, fb1 - (fb0 + (int)blockIdx.x * 4));
  if (blockIdx.z == 0) gram_body<LOSS_TABLE, 0>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, c, g.fb, g.lane, C, N, Fpad, nfb, 1.0, 1.0, g.cfl, g.pz, g.pre, s_cost, nrun, 0, -1, nullptr, 1, 0, ltab);
  else gram_body<LOSS_TABLE, 1>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, c, g.fb, g.lane, C, N, Fpad, nfb, 1.0, 1.0, g.cfl, g.pz, g.pre, s_cost, nrun, 0, -1, nullptr, 1, 0, ltab);
}

// Both roles in one lane: grid.z = 1, one wave per SIMD (all 87 accumulators + temporaries in the 512-register file).
template <int LOSS, bool FAST>
__global__ __launch_bounds__(256) void k_gram(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                 double* __restrict__ rec0, double* __restrict__ rec1, double* __restrict__ gp0, double* __restrict__ gp1, int C, int N, int Fpad, int nfb, int fb0, int fb1,
                                                 double fs2, double ifs2) {
  __shared__ CamConst s_cam;  // camera intrinsics + pose (R, t, Jr) staged once per workgroup
  __shared__ double s_cost[8];
  GramStart g;
  gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, fb0, fb1);
  if (!g.run) return;
  const int nrun = min(4, fb1 - (fb0 + (int)blockIdx.x * 4));  // wavefronts of this workgroup that have a frame block
  gram_body<LOSS, 2, FAST>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, blockIdx.y, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, nrun);
}

// Point split inside the workgroup (gram_body MODE 3): the workgroup's four wavefronts are 4 / NPW (camera, frame block) items of
// NPW wavefronts each, wavefront part = wave % NPW over the points [N part / NPW, N (part + 1) / NPW).  Dynamic LDS: gram_psplit_lds_bytes.
// ROLE 0: the intrinsics of every camera are held fixed (BASELINE configs[1]; camera block 6 wide): role A alone -- 28 accumulators
// instead of 87, no W rows / U rows / g_c of the intrinsics.
template <int LOSS, bool FAST, int NPW, int ROLE = 2>
__global__ __launch_bounds__(256) void k_gram_psplit(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                        double* __restrict__ rec0, double* __restrict__ rec1, double* __restrict__ gp0, double* __restrict__ gp1, int C, int N, int Fpad, int nfb, int fb0, int fb1,
                                                        double fs2, double ifs2) {
  __shared__ CamConst s_cam;
  __shared__ double s_cost[8];
  extern __shared__ __align__(16) double s_xch[];
  const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) % NPW;
  const int p_lo = (int)(((long long)N * part) / NPW), p_hi = (int)(((long long)N * (part + 1)) / NPW);
  GramStart g;
  gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, fb0, fb1, p_lo, max(p_hi, p_lo + 1), NPW);  // (an empty piece -- fewer points than wavefronts -- still prefetches a valid point)
  if (!g.run) return;
  gram_body<LOSS, ROLE, FAST, 3, NPW>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, blockIdx.y, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, 0, p_lo, p_hi, reinterpret_cast<double2*>(s_xch));
}

// Whole rounds of the wavefront slots fused AND the short last round point-split, in ONE launch (round 4): a 1-D grid whose first
// nf C workgroups are k_gram's (four frame blocks of one camera each, the frame blocks [0, fba)), the rest k_gram_psplit's over
// [fba, nfb).  Workgroups are dispatched in id order: every CU starts on a fused workgroup and takes point-split ones as it
// comes free -- the two-launch form paid a kernel boundary (end-of-kernel write-back, dispatch, a second start-up round trip) in between.
template <int LOSS, bool FAST, int NPW>
__global__ __launch_bounds__(256) void k_gram_mixed(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                       double* __restrict__ rec0, double* __restrict__ rec1, double* __restrict__ gp0, double* __restrict__ gp1, int C, int N, int Fpad, int nfb, int fba,
                                                       int nf, int nt, double fs2, double ifs2) {
  __shared__ CamConst s_cam;
  __shared__ double s_cost[8];
  extern __shared__ __align__(16) double s_xch[];
  const int id = blockIdx.x;
  GramStart g;
  if (id < nf * C) {
    const int c = id / nf, bx = id - c * nf;
    gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, 0, fba, 0, -1, 1, bx, c);
    if (!g.run) return;
    const int nrun = min(4, fba - bx * 4);
    gram_body<LOSS, 2, FAST>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, c, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, nrun);
  } else {
    const int id2 = id - nf * C, c = id2 / nt, bx = id2 - c * nt;
    const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) % NPW;
    const int p_lo = (int)(((long long)N * part) / NPW), p_hi = (int)(((long long)N * (part + 1)) / NPW);
    gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, fba, nfb, p_lo, max(p_hi, p_lo + 1), NPW, bx, c);
    if (!g.run) return;
    gram_body<LOSS, 2, FAST, 3, NPW>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, c, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, 0, p_lo, p_hi, reinterpret_cast<double2*>(s_xch));
  }
}

// Point-chunk tail (gram_body MODE 1 / 2).  k_gram_chunk: grid (frame blocks, C, nchunk) of ONE-wavefront workgroups (with four
// wavefronts per workgroup a tail of 14 frame blocks x 24 cameras x 3 chunks is 288 workgroups on 256 CUs: 32 CUs carry two, i.e.
// two wavefronts per SIMD, and the launch takes twice as long -- measured 96.8 us against 5x us), blockIdx.z = chunk of the board's
// points [z ppc, (z + 1) ppc); raw sums -> chunk[((camera * ntb + tail block) * nchunk + z)][44][64] double2.  k_gram_combine: grid
// (ceil(frame blocks / 4), C): sums the chunks of its (camera, frame block) and finishes it exactly as k_gram does.
template <int LOSS, bool FAST>
__global__ __launch_bounds__(64) void k_gram_chunk(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                       double2* __restrict__ chunk, int C, int N, int Fpad, int nfb, int fb0, int fb1, double fs2, double ifs2, int ppc, int nchunk) {
  __shared__ CamConst s_cam;
  __shared__ double s_cost[8];
  const int p_lo = (int)blockIdx.z * ppc, p_hi = min(N, p_lo + ppc);
  GramStart g;
  gram_start(g, s_cam, obs_t, sl, x0, x1, nullptr, nullptr, nullptr, nullptr, C, N, Fpad, fb0, fb1, p_lo, p_hi);
  if (!g.run) return;
  const int ntb = fb1 - fb0;
  double2* dst = chunk + ((((size_t)blockIdx.y * ntb + (g.fb - fb0)) * nchunk + blockIdx.z) * (kGramRaw / 2)) * 64 + g.lane;
  gram_body<LOSS, 2, FAST, 1>(s_cam, obs_t, obj, g.x, nullptr, nullptr, blockIdx.y, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, 0, p_lo, p_hi, dst);
}
template <int LOSS, bool FAST>
__global__ __launch_bounds__(256) void k_gram_combine(const double2* __restrict__ obs_t, const double* __restrict__ obj, Sel sl, const double* __restrict__ x0, const double* __restrict__ x1,
                                                         double* __restrict__ rec0, double* __restrict__ rec1, double* __restrict__ gp0, double* __restrict__ gp1, const double2* __restrict__ chunk,
                                                         int C, int N, int Fpad, int nfb, int fb0, int fb1, double fs2, double ifs2, int nchunk) {
  __shared__ CamConst s_cam;
  __shared__ double s_cost[8];
  GramStart g;
  gram_start(g, s_cam, obs_t, sl, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, fb0, fb1);
  if (!g.run) return;
  const int nrun = min(4, fb1 - (fb0 + (int)blockIdx.x * 4));
  const int ntb = fb1 - fb0;
  const double2* src = chunk + (((size_t)blockIdx.y * ntb + (g.fb - fb0)) * nchunk * (kGramRaw / 2)) * 64 + g.lane;
  gram_body<LOSS, 2, FAST, 2>(s_cam, obs_t, obj, g.x, g.rec, g.gpart, blockIdx.y, g.fb, g.lane, C, N, Fpad, nfb, fs2, ifs2, g.cfl, g.pz, g.pre, s_cost, nrun, 0, N, const_cast<double2*>(src), nchunk,
                              (size_t)(kGramRaw / 2) * 64);
}

// ---------------------------------------------------------------- k_cost: robust cost only (trial points), optional residual vector
// grid (ceil(nfb/4), C, nch): like k_gram, but the board points are split into nch chunks (blockIdx.z) so that
// ~4 wavefronts per SIMD are in flight -- the loop body is short and latency-bound at one wave per SIMD.
template <int LOSS, bool WRITE_RES>
__global__ __launch_bounds__(256) void k_cost(const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x,
                                              double* __restrict__ cpart, double* __restrict__ res, int C, int F, int N, int Fpad, int nfb, int nch, double fs2, double ifs2,
                                              double fill) {   // fill: what a MISSING scalar's slot of the residual vector gets -- 0 (mcba_residuals), or NaN (the detached vector: its own row mask)
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
  constexpr int PF = 4;  // loads run PF points ahead of their use (see k_gram)
  double2 ring[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) ring[j] = op[(size_t)min(p0 + j, p1 - 1) * Fpad];
  auto point = [&](double2 o2, int p) {
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
      *reinterpret_cast<double2*>(res + (((size_t)c * F + f) * N + p) * 2) = make_double2(vu ? ru : fill, vv ? rv : fill);
    }
  };
  int p = p0;
  for (; p + PF <= p1; p += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      double2 o2 = ring[j];
      ring[j] = op[(size_t)min(p + j + PF, p1 - 1) * Fpad];
      point(o2, p + j);
    }
  }
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (p + j < p1) point(ring[j], p + j);
  double cs = wave_sum(cost), ns = wave_sum(nres);
  size_t o = 2 * (((size_t)c * nfb + fb) * nch + ch);
  if (lane == 0) { cpart[o] = cs; cpart[o + 1] = ns; }
}

// ---------------------------------------------------------------- k_syrk:  [decision] + frame factors + partial  sum_f Yx_f Yx_f^T  on the FP64 matrix cores
// The Schur reduction  S = U - sum_f Y_f Y_f^T,  rhs = sum_f Y_f z_f - g_c  is a genuine GEMM (K = 6 F), so it runs
// on v_mfma_f64_16x16x4_f64.  Yx_f = [Y_f ; z_f^T] is (12C+1) x 6: with z as an extra ROW the right-hand side is
// column 12C of the same product.  Rows are padded to NT*16.
// grid (G, ceil(NP / (4*PPW))), block 256 = 4 wavefronts; the stages of FS frames are dealt out evenly: workgroup g owns sq stages,
// the first sr workgroups one more (6 x 10 000: 1250 stages over 512 workgroups = 226 x 3 + 286 x 2 -- workgroups g and g + 256
// share a CU, so no CU gets more than five stages; a uniform 3 stages per workgroup left 161 CUs with six and 95 with three).
//   0. (DECIDE, single-GPU ticks) what used to be k_sum_trial + the decision: EVERY workgroup sums the trial point's cost /
//      step partials (a few KB, fixed order -> identical in every workgroup) and takes the accept / reject decision on an
//      LDS copy of the LM state; workgroup 0 publishes the new state to a SECOND state buffer (the old one is still being
//      read by workgroups that start later).  Saves a launch whose whole content was a 5 us latency chain.
//   1. what used to be k_frame_factor, per super-stage of <= 32 of the workgroup's frames: all 256 threads sum the frames'
//      (V_cf | g_cf) tile rows over the cameras (thread = (frame, part of the rows); every load in flight at once) into LDS,
//      then the first lanes of wavefront 0 (lane = frame): V_f = sum_c V_cf, D_f = diag(V_f) (Marquardt),
//      L L^T = V_f + lambda D_f, z = L^-1 g_f  ->  LDS (L, 1/diag, z) for the stages below and
//      fbuf[f] = {L(21, diagonal slots 1 / L_ii), z(6), g_f(6), D_f(6), pad} for k_backsub.  The W loads of the first stage
//      are already in flight meanwhile.
//   per stage of FS frames:
//   2. every thread forward-substitutes its IPT (row, frame) items from PREFETCHED registers: y = L_f^-1 w, written
//      to LDS as s_y[row][frame*6 + k]  (row stride 6 FS + 2 doubles: conflict-free ds_read_b64 for the MFMA operands);
//   3. the loads of the NEXT stage (W rows from the wave tiles, 6 coalesced loads per item) are issued,
//      then wavefront w accumulates its PPW output tiles (pair q = w + 4k): per K-step of 4 the A operand of tile
//      row ti is one ds_read_b64 per lane (lane l <- s_y[16 ti + (l & 15)][4 ks + (l >> 4)]); B is the same pattern
//      of tile row tj (B = Y^T).  The global-load latency hides under the matrix-core phase.
// Partials: spart[q][reg 0..3][g][lane]  (C/D layout: col = lane & 15, row = (lane >> 4) + 4 reg): the second stage reads
// each (q, reg) slice as ONE contiguous run of G x 512 B.  fpart[g] = {max |g_f|, #failed factorisations} of the workgroup.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
constexpr int kSyrkSuper = 32;  // frames factorised per super-stage: 32 frames x 8 parts of their V / g_f entries = the 256 threads

template <int PPW, int IPT, bool DECIDE, bool XS, int CW = 12>
__global__ __launch_bounds__(256, PPW <= 4 ? 2 : 1) void k_syrk(Sel sl, SyrkFuse fz, const double* __restrict__ rec0, const double* __restrict__ rec1, double* __restrict__ fbuf, double* __restrict__ fpart,
                                              const int* __restrict__ tile_i, const int* __restrict__ tile_j, double* __restrict__ spart, int C, int F, int Fpad, int NT, int NP, int sq, int sr, int FS,
                                              const double* __restrict__ dscale, int xg, int yg) {
  // Workgroup coordinates: (frame set, tile-pair group).  With more than one pair group (the 16-tile variant: 24 cameras = 3) the
  // groups of one frame set read the SAME W rows; launched as (G, 3) they land on three different XCDs (workgroups go round the
  // XCDs by linear id) and each pulls the rows from HBM: 465 MB per launch by the counters against ~150 MB of rows.  xg > 0: a
  // 1-D launch decoded so that the yg siblings of a frame set have the same id modulo 8 --
  // the same XCD, the same L2 -- and follow each other in dispatch order.
  int bx = blockIdx.x, by = blockIdx.y, gx = gridDim.x;
  if (xg > 0) {
    // whole groups of eight frame sets: set (8 q + xcd), its yg siblings in consecutive slots of that XCD; the xg % 8 sets left over
    // are dealt out one workgroup at a time (their siblings may part) -- an XCD must not get more workgroups than it has CUs, or its
    // last one waits for a whole workgroup to finish (first attempt: 33 on three XCDs, 239 us instead of 139 us)
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, nfull = xg >> 3;
    gx = xg;
    if (slot < nfull * yg) {
      by = slot % yg;
      bx = (slot / yg) * 8 + xcd;
    } else {
      const int e = (slot - nfull * yg) * 8 + xcd;
      bx = 8 * nfull + e / yg;
      by = e % yg;
      if (bx >= xg) return;
    }
  }
  extern __shared__ __align__(16) double lds[];
  __shared__ double s_st[MCBA_LMS];
  __shared__ double s_sum[8];
  __shared__ double s_tw;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // scalar: keeps the tile bookkeeping out of the exec mask
  const bool have_state = sl.lms != nullptr;
  if (have_state && t < MCBA_LMS) s_st[t] = sl.lms[t];
  if (DECIDE && t == MCBA_LMS) s_tw = fz.timeout_word ? *fz.timeout_word : -1.0;  // (in flight with the state: not a round trip of its own before the decision)
  if (DECIDE) {
    // trial scalars: wavefront 0 the robust cost of the trial point (BOTH linearisation buffers are summed -- which one holds
    // the trial point depends on the state, and a dependent round trip costs more than the few KB), wavefronts 1..3 the
    // three back-substitution sums.  Same order of additions in every workgroup: bit-identical decisions.
    const double *pa, *pb;
    int stride = 1, count, inner = 1 << 30;
    size_t outer = 0;
    int dense = 1 << 30;
    if (wave == 0) { pa = fz.cp0; pb = fz.cp1; count = fz.ncp; inner = fz.cinner; outer = fz.couter; stride = fz.cstride; dense = fz.cdense; }
    else { pa = pb = fz.bpart + (wave - 1); stride = 3; count = fz.nbp; }
    const bool two = pa != pb;  // wave-uniform
    double sa = 0.0, sb = 0.0;
    for (int base = 0; base < count; base += 512) {
      double va[8], vb[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + lane + 64 * k;
        const int hi = idx / inner, lo = idx - hi * inner;
        const size_t at = (size_t)hi * outer + (size_t)(lo < dense ? lo * stride : dense * stride + (lo - dense));
        va[k] = idx < count ? pa[at] : 0.0;
        vb[k] = (two && idx < count) ? pb[at] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { sa += va[k]; sb += vb[k]; }
    }
    sa = wave_sum63(sa);
    sb = wave_sum63(sb);
    if (lane == 63) { s_sum[wave] = sa; s_sum[4 + wave] = sb; }
  }
  __syncthreads();
  if (have_state && s_st[MCBA_LM_DONE] != 0.0) {  // terminated: nothing left to do (uniform) -- but the rest of the tick reads
    if (DECIDE && bx == 0 && by == 0 && t < MCBA_LMS) fz.lms_post[t] = s_st[t];  // the state from the second buffer
    return;
  }
  if (DECIDE) {
    if (t == 0) {
      // (a back-substitution workgroup of the previous tick's k_solve_backsub gave up waiting: this tick's trial point is stale)
      const bool stale = s_tw == fz.seq_prev && fz.seq_prev > 0.0;
      if (s_st[MCBA_LM_SKIP] != 0.0 || stale) lm_mark_rebuild(s_st);  // the reduced solve had failed: this tick only rebuilds the system
      else {
        const bool trial1 = ((static_cast<int>(s_st[3]) ^ 1) & 1) != 0;  // the trial linearisation lives in the buffer that is not current
        const double tr[8] = {trial1 ? s_sum[4] : s_sum[0], s_sum[1], s_sum[2], s_sum[3], 0.0, 0.0, 0.0, 0.0};
        s_sum[0] = tr[0];
        DecideArgs da = fz.da;
        da.decide = 2;
        da.lms = s_st;
        lm_decide(tr, da);
      }
    }
    __syncthreads();
    if (bx == 0 && by == 0) {  // publish: later kernels of the tick read the state from here
      if (t < MCBA_LMS) fz.lms_post[t] = s_st[t];
      if (t < 4) fz.trial_out[t] = s_sum[t];
    }
  }
  int sidx;
  double lambda;
  if (have_state) {
    const bool spec = sl.spec && s_st[MCBA_LM_SKIP] == 0.0;
    sidx = (static_cast<int>(s_st[3]) ^ sl.idx ^ (spec ? 1 : 0)) & 1;
    lambda = spec ? lm_spec_lambda(s_st[1], sl.lam, sl.dec) : s_st[1];
  } else {
    sidx = sl.idx;
    lambda = sl.lam;
  }
  const double* __restrict__ rec = sidx ? rec1 : rec0;
  const int n = CW * C, nfb = Fpad >> 6;  // CW = 6: the intrinsics of every camera are held fixed -- the rows of (rho, t) alone: W rows 6..11 of each record
  const int RS = 6 * FS + 2;
  double* s_y = lds;                          // [NT*16][RS]
  double* s_L = lds + (size_t)NT * 16 * RS;   // [kSyrkSuper][34]: L(21) 1/diag(6) z(6) pad, one super-stage of frames; behind it [kSyrkSuper][28] sums

  int qi[PPW], rowa[PPW], rowb[PPW];
  mfma_d4 acc[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    int q = by * (4 * PPW) + wave + 4 * k;
    qi[k] = q < NP ? q : -1;  // a missing pair still multiplies tile (0,0) -- branch-free MFMA loop -- but is never stored
    // tile pair q -> (ti, tj), ti <= tj, rows of NT - ti pairs each (the order of the host's table tile_i / tile_j): computed --
    // q is wave-uniform, a few scalar instructions -- because a load here is a dependent global round trip on the critical
    // path of every workgroup, right after the decision
    int ti = 0, tj = 0;
    if (q < NP) {
      int rem = q;
      while (rem >= NT - ti) { rem -= NT - ti; ++ti; }
      tj = ti + rem;
    }
    rowa[k] = (16 * ti + (lane & 15)) * RS + (lane >> 4);
    rowb[k] = (16 * tj + (lane & 15)) * RS + (lane >> 4);
    acc[k] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  }
  for (int i = t; i < (NT * 16 - (n + 1)) * RS; i += 256) s_y[(size_t)(n + 1) * RS + i] = 0.0;  // padding rows stay zero

  const int f0 = (bx * sq + min(bx, sr)) * FS, f1 = min(F, f0 + (sq + (bx < sr ? 1 : 0)) * FS);
  double wreg[IPT][6];

  // thread = (frame b of the stage, row group): its IPT items are rows r0, r0 + 256/FS, ... of the SAME frame, so that
  // frame's L and 1/diag are read from LDS once per stage instead of once per item
  const int tb = t % FS, tr0 = t / FS, rstep = 256 / FS;
  auto prefetch = [&](int fb) {  // issue every global load of one stage; nothing waits here
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
      int row = tr0 + rstep * it, b = tb, f = fb + b;
      if (row < n) {
        int c = row / CW, lr = row - CW * c + (12 - CW);
        const double2* w2 = reinterpret_cast<const double2*>(rec + ((size_t)c * nfb + (f >> 6)) * (MCBA_REC * 64)) + (size_t)(3 * lr) * 64 + (f & 63);
#pragma unroll
        for (int k = 0; k < 3; ++k) { double2 v = w2[k * 64]; wreg[it][2 * k] = v.x; wreg[it][2 * k + 1] = v.y; }
      }
    }
  };
  // frame factors of one super-stage of ns <= 32 frames.  vsum (all 256 threads): thread (frame t & 31, part t >> 5) sums ITS
  // rows of the (V | g_f) tile -- part p owns the double2 rows k = p and p + 8 of the 14 -- over all cameras, every load in
  // flight at once (one memory round trip instead of one per camera batch), and leaves them in LDS; after the barrier the
  // first ns lanes of wavefront 0 damp, factorise (6x6 Cholesky) and forward-substitute their frame.
  double* s_V = s_L + kSyrkSuper * 34;  // [kSyrkSuper][28]
  double gmax = 0.0, nfail = 0.0;
  auto vsum = [&](int s0, int ns) {
    const int fl = t & 31, part = t >> 5;
    const int f = s0 + min(fl, ns - 1);  // idle threads load a duplicate, store nothing
    const double2* r2 = reinterpret_cast<const double2*>(rec + (size_t)(f >> 6) * (MCBA_REC * 64)) + (size_t)(36 + part) * 64 + (f & 63);
    const size_t cstride = (size_t)nfb * (MCBA_REC * 64 / 2);  // double2 elements between two cameras' tiles
    const bool two = part + 8 < 14;
    double2 a0 = make_double2(0.0, 0.0), a1 = make_double2(0.0, 0.0);
    constexpr int CB = 8;  // cameras per batch (16 loads in flight per thread)
    for (int c0 = 0; c0 < C; c0 += CB) {
      double2 u0[CB], u1[CB];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const size_t o = (size_t)min(c0 + j, C - 1) * cstride;  // clamped duplicate loads are ignored below
        u0[j] = r2[o];
        u1[j] = r2[o + (two ? 8 * 64 : 0)];
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (c0 + j < C) { a0.x += u0[j].x; a0.y += u0[j].y; a1.x += u1[j].x; a1.y += u1[j].y; }
      }
    }
    if (fl < ns) {
      double* v = s_V + fl * 28;
      v[2 * part] = a0.x; v[2 * part + 1] = a0.y;
      if (two) { v[2 * (part + 8)] = a1.x; v[2 * (part + 8) + 1] = a1.y; }
    }
  };
  auto factor = [&](int s0, int ns) {  // wavefront 0, lane = frame s0 + lane
    const bool on = lane < ns;
    double* row = s_L + lane * 34;
    if (lane >= kSyrkSuper) return;
    if (on) {
      double V[28];
#pragma unroll
      for (int k = 0; k < 28; ++k) V[k] = s_V[lane * 28 + k];
      double* gf = V + 21;
      double D[6], dsv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double d = V[tri6(k, k)];
        // Marquardt scaling D = diag(J^T J) (x_scale = 'jac'), or the caller's fixed D = 1 / x_scale^2 (numeric x_scale)
        // (XS is a template parameter because even a never-taken vector load here costs the default kernel 1.5 us: the wait
        // for it is also a wait for the next stage's prefetched W rows -- vmcnt retires in order)
        // XS: the per-parameter entry of `dscale` says what this coordinate is -- > 0: the caller's fixed D = 1 / x_scale^2; 0: Marquardt's
        // diag(J^T J) as without it; < 0: FROZEN (a bound is active on it, mcba_set_frozen): it leaves the system -- its row and column
        // of V_f become the identity's, its gradient entry and (below, in the Y build and in k_backsub) its column of every W block
        // count as zero, so its step is exactly 0 and nothing couples to it
        const double ds = XS ? dscale[(size_t)12 * C + 6 * (size_t)(s0 + lane) + k] : 0.0;
        D[k] = ds > 0.0 ? ds : (d > 0.0 ? d : 1.0);
        dsv[k] = ds;
        V[tri6(k, k)] = d + lambda * D[k];
      }
      double gfm[6];
      unsigned fm = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { gfm[k] = gf[k]; if (XS && dsv[k] < 0.0) fm |= 1u << k; }
      if (XS && fm) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if ((fm >> k) & 1u) {
#pragma unroll
            for (int j = 0; j < 6; ++j) V[j <= k ? tri6(j, k) : tri6(k, j)] = j == k ? 1.0 : 0.0;
            gfm[k] = 0.0;
          }
        }
      }
      double Lp[21], id[6], z[6];
      const bool ok = chol6i(V, Lp);  // diagonal slots: 1 / L_ii
#pragma unroll
      for (int k = 0; k < 6; ++k) id[k] = Lp[k * (k + 1) / 2 + k];
      fwd6(Lp, id, gfm, z);
#pragma unroll
      for (int k = 0; k < 21; ++k) row[k] = Lp[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) { row[21 + k] = id[k]; row[27 + k] = z[k]; gmax = fmax(gmax, fabs(gfm[k])); }
      if (XS) row[33] = (double)fm;
      nfail += ok ? 0.0 : 1.0;
      if (by == 0) {
        double o[40];
#pragma unroll
        for (int k = 0; k < 21; ++k) o[k] = Lp[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { o[21 + k] = z[k]; o[27 + k] = gf[k]; o[33 + k] = D[k]; }   // (the TRUE gradient: a frozen coordinate's entry decides when it is released)
        o[39] = XS ? (double)fm : 0.0;   // which coordinates are frozen: k_backsub drops their column of W^T d_c
        double* fbp = fbuf + (size_t)(s0 + lane) * MCBA_FB;
#pragma unroll
        for (int k = 0; k < 40; k += 2) *reinterpret_cast<double2*>(fbp + k) = make_double2(o[k], o[k + 1]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 34; ++k) row[k] = 0.0;  // frames past the end: Y = 0, z = 0
    }
  };

#define SLAP(i) do { } while (0)
  prefetch(f0);
  for (int s0 = f0; s0 < f1; s0 += kSyrkSuper) {
    const int s1 = min(f1, s0 + kSyrkSuper);
    vsum(s0, s1 - s0);
    __syncthreads();
    SLAP(4);
    if (wave == 0) factor(s0, s1 - s0);
    __syncthreads();
    SLAP(0);
    for (int fb = s0; fb < s1; fb += FS) {
      {
        double Lr[27];
        const double* Lp = s_L + (fb - s0 + tb) * 34;
#pragma unroll
        for (int k = 0; k < 27; ++k) Lr[k] = Lp[k];
        const bool live = fb + tb < F;
        const unsigned fm = XS ? (unsigned)Lp[33] : 0u;   // frozen coordinates of this frame: their column of W counts as zero
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
          const int row = tr0 + rstep * it;
          if (row <= n) {
            double* dst = s_y + (size_t)row * RS + tb * 6;
            if (row < n) {
              double yr[6];
              if (XS && fm) {
#pragma unroll
                for (int k = 0; k < 6; ++k) wreg[it][k] = ((fm >> k) & 1u) ? 0.0 : wreg[it][k];
              }
              fwd6(Lr, Lr + 21, wreg[it], yr);
#pragma unroll
              for (int k = 0; k < 6; ++k) dst[k] = live ? yr[k] : 0.0;
            } else {
#pragma unroll
              for (int k = 0; k < 6; ++k) dst[k] = Lp[27 + k];  // z_f (zero for padding frames)
            }
          }
        }
      }
      __syncthreads();
      SLAP(1);
      if (fb + FS < f1) prefetch(fb + FS);  // flies while the matrix cores work
      // matrix-core phase: K = 6 FS in steps of 4 (an even number of steps, except FS = 2 -- more than 26 cameras --
      // where the third step has no partner).  Operands of step ks+1 are read from LDS while the PPW independent MFMAs
      // of step ks issue back to back (64 cycles each on one SIMD).
      const int nks = (6 * FS) / 4;
      // (Measured: skipping the LDS read of a fragment that a neighbouring pair already holds -- wave-uniform branches
      // between the MFMAs -- costs more than the bandwidth it saves: 55k vs 33k cycles for this phase.  Keep it branch-free.)
      double a0[PPW], b0[PPW], a1[PPW], b1[PPW];
#pragma unroll
      for (int k = 0; k < PPW; ++k) { a0[k] = s_y[rowa[k]]; b0[k] = s_y[rowb[k]]; }
      for (int ks = 0; ks + 1 < nks; ks += 2) {
#pragma unroll
        for (int k = 0; k < PPW; ++k) { a1[k] = s_y[rowa[k] + 4 * ks + 4]; b1[k] = s_y[rowb[k] + 4 * ks + 4]; }
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
        if (ks + 2 < nks) {
#pragma unroll
          for (int k = 0; k < PPW; ++k) { a0[k] = s_y[rowa[k] + 4 * ks + 8]; b0[k] = s_y[rowb[k] + 4 * ks + 8]; }
        }
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k], b1[k], acc[k], 0, 0, 0);
      }
      if (nks & 1) {  // FS = 2: the last step's operands are already in a0 / b0
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
      }
      SLAP(2);
      __syncthreads();
      SLAP(3);
    }
  }
  if (wave == 0 && by == 0) {
    const double wm = wave_max(gmax), wn = wave_sum63(nfail);
    if (lane == 63) { fpart[2 * bx] = wm; fpart[2 * bx + 1] = wn; }
  }
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    if (qi[k] >= 0) {
      const size_t G = gx;
      double* o = spart + ((size_t)(4 * qi[k]) * G + bx) * 64 + lane;
      o[0] = acc[k][0]; o[G * 64] = acc[k][1]; o[2 * G * 64] = acc[k][2]; o[3 * G * 64] = acc[k][3];
    }
  }
}

// ---------------------------------------------------------------- k_reduce_system: fixed-order second stage
// Sums the per-workgroup partials of k_syrk and the per-wavefront partials of k_gram into the reduce buffer (layout in
// include/mcba.h) with coalesced reads and a FIXED summation order (bit-reproducible; no FP64 atomics anywhere).
//   blocks [0, 16 NP): one per (tile pair q, accumulator register reg, row rr) = 16 elements that are 128 contiguous bytes in
//       every k_syrk partial (240 blocks at 6 cameras: every CU pulls its share of the 16 MB; one block per (q, reg) left the
//       whole read to 60 CUs: 10.3 -> 7.0 us).  Wave s of 16 sums partials g = s, s+16, ... (all loads in flight: each lane a
//       quarter of the slice's rows), LDS, then wave 0 adds the 16 slices in order and the four row groups.  Elements that fall
//       on a camera's diagonal block also need U_c (and column 12C needs g_c): those sums over the frame blocks are contiguous
//       runs of gpart[camera][k][frame block] -- one wavefront task each.  Off-diagonal tiles are mirrored on write.
//   blocks [16 NP, ...): diag(U), g_c and the 16 scalars, one wavefront task per output.
__device__ __forceinline__ double run_sum(const double* __restrict__ p, int count, int lane) {  // sum of a contiguous run, result in lane 63
  double s = 0.0;
  for (int base = 0; base < count; base += 256) {
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { int i = base + lane + 64 * k; v[k] = i < count ? p[i] : 0.0; }
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  return wave_sum63(s);
}

template <int RR>
__global__ __launch_bounds__(1024) void k_reduce_system(Sel sl, const double* __restrict__ gp0, const double* __restrict__ gp1, const double* __restrict__ spart, const double* __restrict__ fpart,
                                                        const int* __restrict__ tile_i, const int* __restrict__ tile_j, double* __restrict__ red, int C, int nfb, int G, int NT, int NP,
                                                        int nfblocks, int rank_slot, const double* __restrict__ bpart, int nbp, double* __restrict__ state_copy, int cw,
                                                        const double* __restrict__ timeout_word, double seq_prev) {
  const int n = cw * C, coff = 12 - cw;  // cw = 6: intrinsics held fixed -- row i of the system is parameter coff + i % cw of camera i / cw
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ double s_part[RR][16][64];
  __shared__ double s_u[RR][16];
  const size_t camstride = (size_t)MCBA_GP * nfb;
  constexpr int BPQ = 16 / RR;  // blocks per tile pair
  if ((int)blockIdx.x < BPQ * NP) {
    // RR == 1: one block per (tile pair q, accumulator register reg, row rr of the register's four): 16 elements = one 128-byte
    // segment of every k_syrk partial -- four times as many blocks as one per (q, reg), so that the 16 MB of partials are
    // pulled by (almost) every CU instead of 60 of them (6 cameras: 15 pairs).  RR == 4 (many tile pairs: 24 cameras = 171): one
    // block per (q, reg) does the four rows with all their loads in flight at once -- 684 blocks instead of 2 736, each as long
    // as one of those was (a block is a chain of round trips, and only two 1024-thread blocks fit a CU): 21.6 -> 19.2 us by rocprofv3
    // at 24 x 6 250 x 200.
    // Wavefront s sums slice s (the partials g = s, s + 16, ...): lane (rg, el) = (lane >> 4, lane & 15) takes every fourth row
    // of the slice for element el.
    const int q = blockIdx.x / BPQ, reg = RR == 1 ? (blockIdx.x >> 2) & 3 : blockIdx.x & 3, rr0 = RR == 1 ? blockIdx.x & 3 : 0;
    const int ti = tile_i[q], tj = tile_j[q];
    const int rg = lane >> 4, el = lane & 15;
    // ---- the partials do not depend on the LM state: all of this lane's rows (G <= 512 -> at most 8) go in flight before
    // anything waits for the state, which rides along
    constexpr int NK = RR == 1 ? 8 : 2;  // (RR == 4 is chosen for G <= 128 only: two rows per lane and slice)
    double pv[RR][NK];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const double* p = spart + (size_t)(4 * q + reg) * G * 64 + 16 * (rr0 + r) + el;
#pragma unroll
      for (int k = 0; k < NK; ++k) { const int g = wave + 16 * (rg + 4 * k); pv[r][k] = g < G ? p[(size_t)g * 64] : 0.0; }
    }
    if (!sel_active(sl, false)) return;
    const double* __restrict__ gpart = sel_index(sl) ? gp1 : gp0;
    // ---- U_c / g_c term of element `wave` of each of this block's 16-element rows, if it needs one: one wavefront task each
    // (the loads of all RR rows in flight together)
    const double* up[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      up[r] = nullptr;
      const int row = 16 * ti + (rr0 + r) + 4 * reg, col = 16 * tj + wave;
      if (row < n && col < n && row / cw == col / cw) {
        int cam = row / cw, li = row - cw * cam + coff, lj = col - cw * cam + coff;
        int a = li <= lj ? li : lj, b2 = li <= lj ? lj : li;
        up[r] = gpart + cam * camstride + (size_t)tri12(a, b2) * nfb;
      } else if (col == n && row < n) {
        int cam = row / cw, li = row - cw * cam + coff;
        up[r] = gpart + cam * camstride + (size_t)(78 + li) * nfb;
      }
    }
    double us[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) us[r] = 0.0;
    for (int base = 0; base < nfb; base += 256) {
      double w[RR][4];
#pragma unroll
      for (int r = 0; r < RR; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + lane + 64 * k; w[r][k] = (up[r] && i < nfb) ? up[r][i] : 0.0; }
      }
#pragma unroll
      for (int r = 0; r < RR; ++r) us[r] += (w[r][0] + w[r][1]) + (w[r][2] + w[r][3]);
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const double u = wave_sum63(us[r]);
      if (lane == 63) s_u[r][wave] = u;
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < NK; ++k) sum += pv[r][k];
      s_part[r][wave][lane] = sum;
    }
    __syncthreads();
    if (wave < RR) {
      const int r = wave;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += s_part[r][k][lane];  // the 16 slices in order, per (row group, element)
      const double v = (__shfl(t, el, 64) + __shfl(t, el + 16, 64)) + (__shfl(t, el + 32, 64) + __shfl(t, el + 48, 64));  // the four row groups
      if (lane < 16) {
        const int row = 16 * ti + (rr0 + r) + 4 * reg, col = 16 * tj + el;
        if (row < n && col < n) {
          double out = s_u[r][el] - v;  // S0 = blockdiag(U) - sum Y Y^T
          red[(size_t)row * n + col] = out;
          if (ti != tj) red[(size_t)col * n + row] = out;
        } else if (col == n && row < n) {
          red[(size_t)n * n + row] = v - s_u[r][el];  // rhs = sum Y z - g_c
        }
      }
    }
    return;
  }
  // ---- diag(U), g_c, scalars: task id per wavefront
  if (!sel_active(sl, false)) return;
  const double* __restrict__ gpart = sel_index(sl) ? gp1 : gp0;
  const int task = ((int)blockIdx.x - BPQ * NP) * 16 + wave;
  double* tail = red + (size_t)n * n + n;
  if (task < n) {  // diag U
    int cam = task / cw, l = task - cw * cam + coff;
    double v = run_sum(gpart + cam * camstride + (size_t)tri12(l, l) * nfb, nfb, lane);
    if (lane == 63) tail[task] = v;
  } else if (task < 2 * n) {  // g_c
    int jj = task - n, cam = jj / cw, l = jj - cw * cam + coff;
    double v = run_sum(gpart + cam * camstride + (size_t)(78 + l) * nfb, nfb, lane);
    if (lane == 63) tail[n + jj] = v;
  } else if (task < 2 * n + 16) {
    int jj = task - 2 * n;
    double v = 0.0;
    if (jj == 0 || jj == 1) {  // cost, (camera, frame) pairs with data: one pass over all C x nfb per-wavefront sums, 8 loads in
      double a = 0.0;          // flight per lane (a run_sum per camera would be C dependent round trips of ~2 us each)
      const int total = C * nfb;
      for (int base = 0; base < total; base += 512) {
        double w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = base + lane + 64 * k;
          const int cam = i / nfb, fbk = i - cam * nfb;
          w[k] = i < total ? gpart[cam * camstride + (size_t)(90 + jj) * nfb + fbk] : 0.0;
        }
        a += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      }
      v = wave_sum63(a);
    } else if (jj == 2) {
      double a = 0.0;
      for (int k = lane; k < nfblocks; k += 64) a += fpart[2 * k + 1];
      v = wave_sum63(a);
    } else if (jj == 4 + rank_slot) {
      double a = 0.0;
      for (int k = lane; k < nfblocks; k += 64) a = fmax(a, fpart[2 * k]);
      v = wave_max(a);
    }
    if (lane == 63) tail[2 * n + jj] = v;
  } else if (bpart && task < 2 * n + 24) {
    // speculative (frame-sharded) ticks: the trial point's scalars [cost, pred_f, |d_f|^2, |x_f|^2, #pairs, stale, 0, 0] for the
    // all-reduce that follows -- the reduction is built from the trial linearisation, so its cost and pair count are the
    // trial point's; what used to be a launch of its own (k_sum_trial) is eight more wavefront tasks here.
    // stale: 1 if a back-substitution workgroup of THIS shard's previous k_solve_backsub gave up waiting for its solve (the trial
    // point is then stale here).  Summed over the shards by the collective, so that every shard discards the tick -- a shard that
    // rebuilt on its own would part from the others' decisions, and from then on from their sequence of collectives.
    const int jj = task - (2 * n + 16);
    double a = 0.0;
    if (jj == 5 && lane == 0 && timeout_word && seq_prev > 0.0 && *timeout_word == seq_prev) a = 1.0;
    if (jj == 0 || jj == 4) {
      const int total = C * nfb, kk = jj == 0 ? 90 : 91;
      for (int base = 0; base < total; base += 512) {
        double w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = base + lane + 64 * k;
          const int cam = i / nfb, fbk = i - cam * nfb;
          w[k] = i < total ? gpart[cam * camstride + (size_t)kk * nfb + fbk] : 0.0;
        }
        a += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      }
    } else if (jj >= 1 && jj <= 3) {
      for (int base = 0; base < nbp; base += 512) {
        double w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = base + lane + 64 * k; w[k] = i < nbp ? bpart[3 * i + (jj - 1)] : 0.0; }
        a += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      }
    }
    const double v = wave_sum63(a);
    if (lane == 63) tail[2 * n + 16 + jj] = v;
  } else if (bpart && state_copy && task == 2 * n + 24) {
    // ... and a copy of the LM state as it stands BEFORE the decision: k_solve_backsub's back-substitution workgroups read it
    // while the solve of the same launch is already rewriting the original
    if (lane < MCBA_LMS) state_copy[lane] = sl.lms[lane];
  }
}

// ---------------------------------------------------------------- k_backsub: frame steps + trial parameters (body: mcba_backsub.h)
template <class DcSrc, int CW>
__global__ __launch_bounds__(64 * kBacksubWaves) void k_backsub(Sel sl, const double* __restrict__ rec0, const double* __restrict__ rec1, const double* __restrict__ fbuf, const DcSrc dcs,
                                                                double* __restrict__ x0, double* __restrict__ x1, double* __restrict__ bpart, int C, int F, int Fpad) {
  __shared__ double s_t[kBacksubWaves][6][64];
  backsub_body<DcSrc, CW>(sl, rec0, rec1, fbuf, dcs, x0, x1, bpart, C, F, Fpad, (int)blockIdx.x, (int)(blockDim.x >> 6), s_t, nullptr);
}

// trial scalars: [cost, pred_f, dn2_f, xn2_f, n_residuals, 0, 0, 0].  One block of 512 threads: wavefront w
// produces scalar w; its lanes stride over the partials with 8 loads in flight each, then a fixed DPP tree.
// Cost partials come either from k_cost (stride 2: cost, n_residuals) or from k_gram's per-wave sums
// (stride MCBA_GP, entries 90 / 91) when the trial point was linearised speculatively.
// state (lms): 0 cost  1 lambda  2 nu  3 sel (current slot / linearisation)  4 accepted  5 cost_new  6 pred  7 ratio
//        8 step_norm  9 x_norm  10 dF.   pred_cam = d_c^T (lam D_c d_c - g_c), dcn2 = |d_c|^2, xcn2 = |x_c|^2 come from
// the host (it solved the camera system).  Nielsen's update on acceptance, doubling growth on rejection -- identical
// to solver.LevenbergMarquardt.iterate.
// Cost partials: element idx of `ncp` lives at (idx / cinner) * couter + (idx % cinner) * cstride  (k_cost: cinner = ncp,
// cstride = 2; k_gram: per camera a contiguous run of nfb values, cinner = nfb, couter = 92 nfb, cstride = 1).
__global__ __launch_bounds__(512) void k_sum_trial(Sel sl, const double* __restrict__ cp0, const double* __restrict__ cp1, int cstride, int cinner, size_t couter, int ncp, const double* __restrict__ bpart, int nbp, double* __restrict__ out, DecideArgs da) {
  // Which buffer holds the trial point's partials depends on the LM state -- a dependent round trip of ~2 us for a kernel that
  // does 4 us of everything.  The partials are tiny: BOTH candidates are summed, with the state read in flight next to
  // them, and the right sum is picked afterwards.
  double st3 = 0.0, st14 = 0.0, st15 = 0.0;
  if (sl.lms) { st3 = sl.lms[3]; st14 = sl.lms[MCBA_LM_SKIP]; st15 = sl.lms[MCBA_LM_DONE]; }
  LmPre pre = {};
  if (da.decide && threadIdx.x == 0) lm_prefetch(da.lms, pre);  // the decision's inputs travel with the first batch of loads
  __shared__ double s_out[8];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const double *pa = nullptr, *pb = nullptr;
  int stride = 1, count = 0, inner = 1 << 30;
  size_t outer = 0;
  if (w == 0) { pa = cp0; pb = cp1; stride = cstride; count = ncp; inner = cinner; outer = couter; }
  else if (w >= 1 && w <= 3 && bpart) { pa = pb = bpart + (w - 1); stride = 3; count = nbp; }
  else if (w == 4) {  // n_residuals / pairs with data
    const size_t off = cstride == 2 ? 1 : (size_t)cinner;
    pa = cp0 + off; pb = cp1 + off; stride = cstride; count = ncp; inner = cinner; outer = couter;
  }
  const bool two = pa != pb;  // wave-uniform
  double sa = 0.0, sb = 0.0;
  for (int base = 0; base < count; base += 512) {
    double va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int idx = base + l + 64 * k;
      int hi = idx / inner, lo = idx - hi * inner;
      const size_t at = (size_t)hi * outer + (size_t)lo * stride;
      va[k] = idx < count ? pa[at] : 0.0;
      vb[k] = (two && idx < count) ? pb[at] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { sa += va[k]; sb += vb[k]; }
  }
  const bool active = !sl.lms || (st15 == 0.0 && st14 == 0.0);  // sel_active(sl, true)
  if (!active) {
    if (da.decide && threadIdx.x == 0 && st15 == 0.0) lm_mark_rebuild(da.lms);
    return;
  }
  const int sidx = sl.lms ? ((static_cast<int>(st3) ^ sl.idx) & 1) : sl.idx;
  double s = (two && sidx) ? sb : sa;
  s = wave_sum63(s);
  if (l == 63) { out[w] = s; s_out[w] = s; }
  if (da.decide) {  // single-rank runs: the accept/reject decision rides on the same launch
    __syncthreads();
    if (threadIdx.x == 0) lm_decide(s_out, da, pre);
  }
}

// stand-alone decision (frame-sharded runs: the trial scalars are all-reduced between k_sum_trial and this)
__global__ void k_decide(const double* __restrict__ trial8, DecideArgs da) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || da.lms[MCBA_LM_DONE] != 0.0) return;
  if (da.lms[MCBA_LM_SKIP] != 0.0) lm_mark_rebuild(da.lms);
  else lm_decide(trial8, da);
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
      typedef double nt_d2 __attribute__((ext_vector_type(2)));
      nt_d2 v = {s_rows[pp * 37 + kk], s_rows[pp * 37 + kk + 1]};
      __builtin_nontemporal_store(v, reinterpret_cast<nt_d2*>(out) + j);  // 985 MB written once and not read back by the GPU: non-temporal
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
  k_transpose_obs<<<dim3(Fpad / 64, C, (N + 15) / 16), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(raw), reinterpret_cast<double2*>(obs_t), C, F, N, Fpad);
}

// slots: wavefront slots of the handle's device at one wavefront per SIMD (4 x compute units; derive_geometry keeps it in the handle --
// not a process-wide value: handles on devices of different sizes may be driven from different threads)
int gram_round_blocks(int C, int nfb, int slots) { return std::min(nfb, (((C * nfb) / slots) * slots / C) & ~3); }
size_t gram_psplit_lds_bytes(int npw, int cw) {
  size_t b = (size_t)gram_xch_doubles(npw == 4 ? 4 : 2, cw != 6) * sizeof(double);
  return b;
}

// Measurement aid: the next fused k_gram launch of this thread carries these two events ON ITS DISPATCH (hipExtLaunchKernelGGL): the
// kernel's own begin / end timestamps.  Launch variants other than the fused kernel ignore them (the caller then brackets as usual).
static thread_local hipEvent_t t_ext_start = nullptr, t_ext_stop = nullptr;
void gram_time_next_launch(hipEvent_t start, hipEvent_t stop) { t_ext_start = start; t_ext_stop = stop; }
bool gram_time_pending() { return t_ext_start != nullptr; }

void launch_gram(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, Sel s, const double* x0, const double* x1, double* rec0, double* rec1, double* gp0, double* gp1, int C, int N, int Fpad, int split,
                 int planar, double* chunk, int nchunk, int npw, int cw, int slots, const double* ltab) {
  const int nfb = Fpad / 64;
  dim3 block(256);
  if (loss == LOSS_TABLE) {   // (the caller's tabulated loss: one launch variant, whatever the shape; the intrinsics held fixed = role A alone)
    k_gram_table<<<dim3((nfb + 3) / 4, C, cw == 6 ? 1 : 2), block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, 0, nfb, reinterpret_cast<const double2*>(ltab));
    return;
  }
  const double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  const double2* o2 = reinterpret_cast<const double2*>(obs_t);
  auto fused = [&](int fb0, int fb1) {
    dim3 grid((fb1 - fb0 + 3) / 4, C, 1);
    if (t_ext_start && t_ext_stop) {
      // measurement (gram_time_next_launch): the events ride on the dispatch itself -- they get the kernel's own begin / end
      // timestamps, what rocprofv3 reports, not the arrival times of barrier packets around it
      hipEvent_t ea = t_ext_start, eb = t_ext_stop;
      t_ext_start = t_ext_stop = nullptr;
      if (planar && f_scale == 1.0) {
        DISPATCH_LOSS(loss, (hipExtLaunchKernelGGL((k_gram<L, true>), grid, block, 0, st, ea, eb, 0, o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fb0, fb1, fs2, ifs2)));
      } else {
        DISPATCH_LOSS(loss, (hipExtLaunchKernelGGL((k_gram<L, false>), grid, block, 0, st, ea, eb, 0, o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fb0, fb1, fs2, ifs2)));
      }
      return;
    }
    if (planar && f_scale == 1.0) {
      DISPATCH_LOSS(loss, (k_gram<L, true><<<grid, block, 0, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fb0, fb1, fs2, ifs2)));
    } else {
      DISPATCH_LOSS(loss, (k_gram<L, false><<<grid, block, 0, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fb0, fb1, fs2, ifs2)));
    }
  };
  auto roles = [&](int fb0, int fb1) {
    dim3 grid((fb1 - fb0 + 3) / 4, C, 2);
    DISPATCH_LOSS(loss, (k_gram_split<L><<<grid, block, 0, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fb0, fb1, fs2, ifs2)));
  };
  // point split inside the workgroup: npw = 4 / 2 wavefronts per (camera, frame block), 1 / 2 of them per workgroup
  auto psplit = [&](int fb0, int fb1) {
    const int per = npw == 4 ? 1 : 2;
    dim3 grid((fb1 - fb0 + per - 1) / per, C, 1);
    const size_t lds = gram_psplit_lds_bytes(npw == 4 ? 4 : 2, cw);
    const bool fast = planar && f_scale == 1.0;
#define PS_GO(FASTV, NPWV, ROLEV) DISPATCH_LOSS(loss, (k_gram_psplit<L, FASTV, NPWV, ROLEV><<<grid, block, lds, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fb0, fb1, fs2, ifs2)))
    if (cw == 6) {
      if (npw == 4) {
        if (fast) { PS_GO(true, 4, 0); } else { PS_GO(false, 4, 0); }
      } else {
        if (fast) { PS_GO(true, 2, 0); } else { PS_GO(false, 2, 0); }
      }
    } else if (npw == 4) {
      if (fast) { PS_GO(true, 4, 2); } else { PS_GO(false, 4, 2); }
    } else {
      if (fast) { PS_GO(true, 2, 2); } else { PS_GO(false, 2, 2); }
    }
#undef PS_GO
  };
  if (cw == 6) {
    // intrinsics held fixed (camera block 6 wide): role A alone -- the point split for shards of at most half a round of the wavefront
    // slots, otherwise the role-A half of the split-role kernel (grid.z = 1: two wavefronts per SIMD)
    if (split == 4) { psplit(0, nfb); return; }
    dim3 grid((nfb + 3) / 4, C, 1);
    DISPATCH_LOSS(loss, (k_gram_split<L><<<grid, block, 0, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, 0, nfb, fs2, ifs2)));
    return;
  }
  if (split == 4) { psplit(0, nfb); return; }
  if (split == 5) {  // whole rounds of the wavefront slots fused, the short last round point-split
    const int fba5 = gram_round_blocks(C, nfb, slots);
    if (fba5 > 0 && fba5 < nfb) {
      static const bool two_launches = [] { const char* e = getenv("MCBA_GRAM_MIXED"); return e && atoi(e) == 0; }();  // (0: the two-launch form, for A/B)
      if (two_launches) { fused(0, fba5); psplit(fba5, nfb); return; }
      const int per = npw == 4 ? 1 : 2, nf = (fba5 + 3) / 4, nt = (nfb - fba5 + per - 1) / per;
      const size_t lds = gram_psplit_lds_bytes(npw == 4 ? 4 : 2, 12);
      const bool fast = planar && f_scale == 1.0;
      dim3 grid((nf + nt) * C);
#define MX_GO(FASTV, NPWV) DISPATCH_LOSS(loss, (k_gram_mixed<L, FASTV, NPWV><<<grid, block, lds, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, C, N, Fpad, nfb, fba5, nf, nt, fs2, ifs2)))
      if (npw == 4) {
        if (fast) { MX_GO(true, 4); } else { MX_GO(false, 4); }
      } else {
        if (fast) { MX_GO(true, 2); } else { MX_GO(false, 2); }
      }
#undef MX_GO
    } else fused(0, nfb);
    return;
  }
  if (split == 1) { roles(0, nfb); return; }
  if (split == 3) {
    // whole rounds of the 1 024 wavefront slots fused; the last, short round as POINT CHUNKS: nchunk wavefronts per (camera, frame
    // block), each over 1 / nchunk of the board's points (ppc a multiple of the loop's four), + one combine launch
    const int items3 = C * nfb;
    const int fba3 = ((items3 / slots) * slots / C) & ~3;
    if (fba3 > 0 && fba3 < nfb && chunk && nchunk >= 2) {
      fused(0, fba3);
      const int ppc = ((N + nchunk - 1) / nchunk + 3) & ~3;
      const int nch = (N + ppc - 1) / ppc;
      dim3 gridc(nfb - fba3, C, nch), gridm((nfb - fba3 + 3) / 4, C, 1);
      double2* ck = reinterpret_cast<double2*>(chunk);
      const bool fast = planar && f_scale == 1.0;
      if (fast) {
        DISPATCH_LOSS(loss, (k_gram_chunk<L, true><<<gridc, dim3(64), 0, st>>>(o2, obj, s, x0, x1, ck, C, N, Fpad, nfb, fba3, nfb, fs2, ifs2, ppc, nch)));
        DISPATCH_LOSS(loss, (k_gram_combine<L, true><<<gridm, block, 0, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, ck, C, N, Fpad, nfb, fba3, nfb, fs2, ifs2, nch)));
      } else {
        DISPATCH_LOSS(loss, (k_gram_chunk<L, false><<<gridc, dim3(64), 0, st>>>(o2, obj, s, x0, x1, ck, C, N, Fpad, nfb, fba3, nfb, fs2, ifs2, ppc, nch)));
        DISPATCH_LOSS(loss, (k_gram_combine<L, false><<<gridm, block, 0, st>>>(o2, obj, s, x0, x1, rec0, rec1, gp0, gp1, ck, C, N, Fpad, nfb, fba3, nfb, fs2, ifs2, nch)));
      }
      return;
    }
    fused(0, nfb);
    return;
  }
  // split == 2: whole rounds of the 1024 wavefront slots with the fused variant, the (short) last round with the split
  // roles -- their wavefronts are lighter, so a tail of r items costs ~0.55 of a fused pass instead of a whole one
  const int items = C * nfb;
  const int fba = split == 2 ? (((items / slots) * slots / C) & ~3) : nfb;
  if (split == 2 && fba > 0 && fba < nfb) {
    fused(0, fba);
    roles(fba, nfb);
  } else {
    fused(0, nfb);
  }
}

size_t gram_chunk_doubles(int C, int nfb, int nchunk, int slots) {
  const int items = C * nfb, fba = ((items / slots) * slots / C) & ~3;
  return (size_t)C * (nfb - fba) * nchunk * kGramRaw * 64;
}

void launch_cost(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, const double* x, double* cpart, double* res, int C, int F, int N, int Fpad, int nch, double fill) {
  int nfb = Fpad / 64;
  dim3 grid((nfb + 3) / 4, C, nch), block(256);
  double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  if (loss == LOSS_TABLE) loss = LOSS_LINEAR;   // (residuals are what this launch is for then: the cost of a tabulated loss is the caller's to evaluate)
  if (res) {
    DISPATCH_LOSS(loss, (k_cost<L, true><<<grid, block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, cpart, res, C, F, N, Fpad, nfb, nch, fs2, ifs2, fill)));
  } else {
    DISPATCH_LOSS(loss, (k_cost<L, false><<<grid, block, 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, cpart, res, C, F, N, Fpad, nfb, nch, fs2, ifs2, fill)));
  }
}

size_t syrk_lds_bytes(int C, int FS, int cw) {
  int NT = (cw * C + 1 + 15) / 16;
  return ((size_t)NT * 16 * (6 * FS + 2) + (size_t)kSyrkSuper * (34 + 28)) * sizeof(double);
}

#define SYRK_IPT 5  // (12C+1)*FS <= 256*SYRK_IPT is guaranteed by the choice of FS in mcba_create
#define SYRK_IPT_SMALL 3

static bool g_syrk_xcd_remap = [] { const char* e = getenv("MCBA_SYRK_XCD"); return !e || atoi(e) != 0; }();  // (0: the plain (G, groups) launch, for A/B)
void launch_syrk(hipStream_t st, Sel s, const SyrkFuse& fz, const double* rec0, const double* rec1, double* fbuf, double* fpart, const int* tile_i, const int* tile_j, double* spart, int C, int F, int Fpad, int NT, int NP, int G, int sq, int sr, int FS, int ppw,
                 const double* dscale, int cw) {
  size_t lds = syrk_lds_bytes(C, FS, cw);
#define SYRK_GO(PPW, IPT, GY, CWV)                                                                                                                              \
  do {                                                                                                                                                          \
    const bool remap = (GY) > 1 && g_syrk_xcd_remap;                                                                                                            \
    dim3 grid(remap ? 8 * ((G / 8) * (GY) + ((G % 8) * (GY) + 7) / 8) : G, remap ? 1 : (GY));                                                                   \
    const int xg = remap ? G : 0, yg = remap ? (GY) : 0;                                                                                                        \
    if (fz.decide && dscale) k_syrk<PPW, IPT, true, true, CWV><<<grid, dim3(256), lds, st>>>(s, fz, rec0, rec1, fbuf, fpart, tile_i, tile_j, spart, C, F, Fpad, NT, NP, sq, sr, FS, dscale, xg, yg);    \
    else if (fz.decide) k_syrk<PPW, IPT, true, false, CWV><<<grid, dim3(256), lds, st>>>(s, fz, rec0, rec1, fbuf, fpart, tile_i, tile_j, spart, C, F, Fpad, NT, NP, sq, sr, FS, dscale, xg, yg);      \
    else if (dscale) k_syrk<PPW, IPT, false, true, CWV><<<grid, dim3(256), lds, st>>>(s, fz, rec0, rec1, fbuf, fpart, tile_i, tile_j, spart, C, F, Fpad, NT, NP, sq, sr, FS, dscale, xg, yg);         \
    else k_syrk<PPW, IPT, false, false, CWV><<<grid, dim3(256), lds, st>>>(s, fz, rec0, rec1, fbuf, fpart, tile_i, tile_j, spart, C, F, Fpad, NT, NP, sq, sr, FS, dscale, xg, yg);                    \
  } while (0)
  // items per thread: (12C + 1) rows x FS frames over 256 threads -- 3 is enough up to 7 cameras at 8 frames per stage
  // (fewer prefetch registers: the kernel stays within 256 registers, two workgroups per CU, without scratch)
  const bool small = (cw * C + 1) * FS <= 256 * SYRK_IPT_SMALL;
  if (cw == 6) {  // intrinsics held fixed: rows of (rho, t) alone; mcba_set_camera_block admits it only where the 4-tile variant serves (<= 64 tile pairs)
    if (small) SYRK_GO(4, SYRK_IPT_SMALL, (NP + 15) / 16, 6);
    else SYRK_GO(4, SYRK_IPT, (NP + 15) / 16, 6);
  } else if (ppw <= 4) {
    if (small) SYRK_GO(4, SYRK_IPT_SMALL, (NP + 15) / 16, 12);
    else SYRK_GO(4, SYRK_IPT, (NP + 15) / 16, 12);
  } else {
    SYRK_GO(16, SYRK_IPT, (NP + 63) / 64, 12);
  }
#undef SYRK_GO
}

int syrk_items_per_thread() { return SYRK_IPT; }

void launch_reduce_system(hipStream_t st, Sel s, const double* gp0, const double* gp1, const double* spart, const double* fpart, const int* tile_i, const int* tile_j, double* red, int C, int nfb, int G, int NT, int NP, int nfblocks, int rank_slot,
                          const double* bpart, int nbp, double* state_copy, int cw, const double* timeout_word, double seq_prev) {
  int n = cw * C;
  int tail_blocks = (2 * n + 16 + (bpart ? 9 : 0) + 15) / 16;
  // many tile pairs and few partials per pair (the 16-tile k_syrk: one workgroup per CU): one block per (pair, register) instead of four
  if (NP >= 64 && G <= 128) k_reduce_system<4><<<dim3(4 * NP + tail_blocks), dim3(1024), 0, st>>>(s, gp0, gp1, spart, fpart, tile_i, tile_j, red, C, nfb, G, NT, NP, nfblocks, rank_slot, bpart, nbp, state_copy, cw, timeout_word, seq_prev);
  else k_reduce_system<1><<<dim3(16 * NP + tail_blocks), dim3(1024), 0, st>>>(s, gp0, gp1, spart, fpart, tile_i, tile_j, red, C, nfb, G, NT, NP, nfblocks, rank_slot, bpart, nbp, state_copy, cw, timeout_word, seq_prev);
}

void launch_backsub(hipStream_t st, Sel s, const double* rec0, const double* rec1, const double* fbuf, const CamStep& dc, double* x0, double* x1, double* bpart, int C, int F, int Fpad, int cw) {
  if (cw == 6) k_backsub<CamStep, 6><<<dim3(Fpad / 64), dim3(64 * std::min(C, kBacksubWaves)), 0, st>>>(s, rec0, rec1, fbuf, dc, x0, x1, bpart, C, F, Fpad);
  else k_backsub<CamStep, 12><<<dim3(Fpad / 64), dim3(64 * std::min(C, kBacksubWaves)), 0, st>>>(s, rec0, rec1, fbuf, dc, x0, x1, bpart, C, F, Fpad);
}
void launch_backsub_dev(hipStream_t st, Sel s, const double* rec0, const double* rec1, const double* fbuf, const double* dc_dev, double* x0, double* x1, double* bpart, int C, int F, int Fpad, int cw) {
  if (cw == 6) k_backsub<DevStep, 6><<<dim3(Fpad / 64), dim3(64 * std::min(C, kBacksubWaves)), 0, st>>>(s, rec0, rec1, fbuf, DevStep{dc_dev}, x0, x1, bpart, C, F, Fpad);
  else k_backsub<DevStep, 12><<<dim3(Fpad / 64), dim3(64 * std::min(C, kBacksubWaves)), 0, st>>>(s, rec0, rec1, fbuf, DevStep{dc_dev}, x0, x1, bpart, C, F, Fpad);
}

void launch_sum_trial(hipStream_t st, Sel s, const double* cp0, const double* cp1, int cstride, int cinner, size_t couter, int ncp, const double* bpart, int nbp, double* out, DecideArgs da) {
  k_sum_trial<<<dim3(1), dim3(512), 0, st>>>(s, cp0, cp1, cstride, cinner, couter, ncp, bpart, nbp, out, da);
}

// mcba_lm_run: the LM state a fresh solve starts from, written on the device -- its cost is scalar 0 of the reduced system that was
// just built (and all-reduced) at the start point, so no host round trip separates that build from the first solve
__global__ void k_lm_init(const double* __restrict__ red_scal, double* __restrict__ lms, double lam0, double sel, double cfl, double cfl_switch, double* __restrict__ clear8) {
  const int i = threadIdx.x;
  if (clear8 && i >= 56) clear8[i - 56] = 0.0;   // (the release words of the fused back-substitution: what a memset of their own did)
  if (i >= MCBA_LMS) return;
  double v = 0.0;
  if (i == 0) v = red_scal[0];
  else if (i == 1) v = lam0;
  else if (i == 2) v = 2.0;
  else if (i == 3) v = sel;
  else if (i == MCBA_LM_CFL) v = cfl;
  else if (i == MCBA_LM_CFL_SWITCH) v = cfl_switch;
  lms[i] = v;
}
void launch_lm_init(hipStream_t st, const double* red_scal, double* lms, double lam0, int sel, double cfl, double cfl_switch, double* clear8) {
  static_assert(MCBA_LMS <= 56, "threads 56..63 of k_lm_init clear the eight release words");
  k_lm_init<<<dim3(1), dim3(64), 0, st>>>(red_scal, lms, lam0, (double)sel, cfl, cfl_switch, clear8);
}

// mcba_lm_result: [x (12C + 6F) | gradient (12C + 6F)] of the current point in one buffer -- the camera gradient g_c of the reduced system
// scattered to the parameter layout (zero where a parameter is held fixed), the frame gradients from the frame records
__global__ __launch_bounds__(256) void k_pack_result(const double* __restrict__ x, const double* __restrict__ gc, const double* __restrict__ fbuf, const unsigned char* __restrict__ fixed, double* __restrict__ out, int C, int F,
                                                     int cw) {
  const size_t nx = (size_t)12 * C + (size_t)6 * F, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nx) return;
  out[i] = x[i];
  double g;
  if (i < (size_t)12 * C) {
    const int c = (int)i / 12, k = (int)i % 12;
    const int row = cw == 12 ? (int)i : (k >= 6 ? 6 * c + (k - 6) : -1);
    g = row >= 0 && !(fixed && fixed[row]) ? gc[row] : 0.0;
  } else {
    const size_t j = i - (size_t)12 * C, f = j / 6, k = j - 6 * f;
    g = fbuf[f * MCBA_FB + 27 + k];
  }
  out[nx + i] = g;
}
void launch_pack_result(hipStream_t st, const double* x, const double* gc, const double* fbuf, const unsigned char* fixed, double* out, int C, int F, int cw) {
  const size_t nx = (size_t)12 * C + (size_t)6 * F;
  k_pack_result<<<dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st>>>(x, gc, fbuf, fixed, out, C, F, cw);
}

void launch_decide(hipStream_t st, const double* trial8, DecideArgs da) {
  k_decide<<<dim3(1), dim3(64), 0, st>>>(trial8, da);
}

void launch_jacobian(hipStream_t st, int loss, double f_scale, const double* obs_raw, const double* obj, const double* x, double* jac, double* res, int C, int F, int N, int Fpad, int robust) {
  if (loss == LOSS_TABLE) { loss = LOSS_LINEAR; robust = 0; }   // (tabulated loss: the unscaled rows; the caller scales them with its own rho)
  double fs2 = f_scale * f_scale, ifs2 = 1.0 / fs2;
  DISPATCH_LOSS(loss, (k_jacobian<L><<<dim3(F, C), dim3(64), 0, st>>>(reinterpret_cast<const double2*>(obs_raw), obj, x, jac, res, C, F, N, Fpad, robust, fs2, ifs2)));
}

int gram_psplit_set_lds_limit() {
  int rc = 0;
#define PS_K(L) reinterpret_cast<const void*>(k_gram_psplit<L, true, 4, 2>), reinterpret_cast<const void*>(k_gram_psplit<L, false, 4, 2>), \
                reinterpret_cast<const void*>(k_gram_psplit<L, true, 2, 2>), reinterpret_cast<const void*>(k_gram_psplit<L, false, 2, 2>)
#define MX_K(L) reinterpret_cast<const void*>(k_gram_mixed<L, true, 4>), reinterpret_cast<const void*>(k_gram_mixed<L, false, 4>), \
                reinterpret_cast<const void*>(k_gram_mixed<L, true, 2>), reinterpret_cast<const void*>(k_gram_mixed<L, false, 2>)
  const void* ks[] = {PS_K(LOSS_LINEAR), PS_K(LOSS_SOFT_L1), PS_K(LOSS_HUBER), PS_K(LOSS_CAUCHY), PS_K(LOSS_ARCTAN),
                      MX_K(LOSS_LINEAR), MX_K(LOSS_SOFT_L1), MX_K(LOSS_HUBER), MX_K(LOSS_CAUCHY), MX_K(LOSS_ARCTAN)};
#undef MX_K
#undef PS_K
  for (const void* k : ks) {
    int r = (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_psplit_lds_bytes(4, 12));
    rc = rc ? rc : r;
  }
  return rc;
}

int syrk_set_lds_limit(size_t bytes) {
  int rc = 0;
#define SYRK_K(P, I, W) reinterpret_cast<const void*>(k_syrk<P, I, false, false, W>), reinterpret_cast<const void*>(k_syrk<P, I, true, false, W>), \
                        reinterpret_cast<const void*>(k_syrk<P, I, false, true, W>), reinterpret_cast<const void*>(k_syrk<P, I, true, true, W>)
  const void* ks[] = {SYRK_K(4, SYRK_IPT, 12), SYRK_K(4, SYRK_IPT_SMALL, 12), SYRK_K(16, SYRK_IPT, 12), SYRK_K(4, SYRK_IPT, 6), SYRK_K(4, SYRK_IPT_SMALL, 6)};
#undef SYRK_K
  for (const void* k : ks) {
    int r = (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    rc = rc ? rc : r;
  }
  return rc;
}

}  // namespace mcba
