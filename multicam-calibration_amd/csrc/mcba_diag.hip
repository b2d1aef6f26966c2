// mcba_diag.hip -- what surrounds the solver on the GPU: the reference's frame pre-filter, frame subsets without a second
// upload, stand-alone undistortion and the numeric core of its reprojection diagnostics.
//
//   k_frame_err      bundle_adjust()'s pre-filter (reference bundle_adjustment.py:265-285): per (camera, frame, point) the
//                    reprojection error |observed - predicted| (NaN where a coordinate is missing), per (camera, frame) its
//                    nan-mean over the board points and the number of complete points.  Same mapping as k_cost: lane = frame,
//                    64 consecutive frames per wavefront, loop over the points -> the means are lane-local.
//   k_sel_hist/pick  exact nan-median of the errors of the selected frames (`5 * np.nanmedian(err)`, :281) without sorting:
//                    radix select on the bit patterns of the (non-negative) doubles, 8 passes of one byte; histogram in LDS,
//                    integer atomics only -> the result is the exact order statistic, bit for bit.
//   k_gather_frames  observations of a frame subset, device to device, for the handle the solver then runs on.
//   k_seen_bits      which observation scalars are present, one bit each in numpy.packbits order: the row selection of the
//                    reference's residual vector and Jacobian (bundle_adjustment.py:68-69, 101) taken from the GPU's own copy.
//   k_undistort      `undistort_points` (geometry.py:328-358): OpenCV's fixed-point iteration for (k1 k2 p1 p2 k3).
//   k_reproj_diag    `plot_residuals` (viz.py:166-186) without the plotting: distortion-free reprojection of the board,
//                    least-squares homography from the undistorted detections to the board plane per (camera, frame),
//                    reprojections mapped through it, distance to the board points.  Lane = (camera, frame).
#include <hip/hip_runtime.h>
#include <mutex>
#include <math.h>
#include <string.h>
#include <algorithm>
#include "mcba_kernels.h"
#include "mcba_math.h"

namespace mcba {

__device__ __forceinline__ double diag_uni(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

// ---------------------------------------------------------------- pre-filter errors
__global__ __launch_bounds__(256) void k_frame_err(const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x, double* __restrict__ err,
                                                   double* __restrict__ mean_cf, double* __restrict__ full_cf, int C, int F, int N, int Fpad, int nfb,
                                                   unsigned long long* __restrict__ clear, unsigned nclear) {
  // (mcba_prefilter: the selection's state is zeroed HERE, by the launch that precedes its first pass anyway -- a hipMemsetAsync of its own is two
  //  fill kernels of ~5 us each between this launch and the next)
  if (clear)
    for (unsigned i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nclear; i += gridDim.x * gridDim.y * 256) clear[i] = 0ull;
  __shared__ CamConst s_cam;
  const int c = blockIdx.y;
  if (threadIdx.x == 0) make_cam_const(x + 12 * c, s_cam);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fb = blockIdx.x * 4 + wave;
  if (fb >= nfb) return;
  const int f = fb * 64 + lane;
  Intr K;
  K.fx = diag_uni(s_cam.fx); K.fy = diag_uni(s_cam.fy); K.cx = diag_uni(s_cam.cx); K.cy = diag_uni(s_cam.cy); K.k1 = diag_uni(s_cam.k1); K.k2 = diag_uni(s_cam.k2);
  double Rc[9], tc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = diag_uni(s_cam.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) tc[i] = diag_uni(s_cam.t[i]);
  const double* pose = x + 12 * C + 6 * (size_t)f;  // padding frames: zeros in x, NaN observations
  double pz[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pz[i] = pose[i];
  PairConst pc;
  {
    double Rf[9];
    rot_only(pz, Rf);
    make_pair_const(Rc, tc, Rf, pz + 3, pc);
  }
  const double2* op = obs_t + (size_t)c * N * Fpad + f;
  double* ep = err + (size_t)c * N * Fpad + f;
  constexpr int PF = 4;
  double2 ring[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) ring[j] = op[(size_t)min(j, N - 1) * Fpad];
  double sum = 0.0, cnt = 0.0, full = 0.0;
  auto point = [&](double2 o2, int p) {
    full += (o2.x == o2.x && o2.y == o2.y) ? 1.0 : 0.0;  // completeness looks at the detections only (bundle_adjustment.py:266)
    const double Xo[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
    double up, vp;
    project_only(K, pc, Xo, up, vp);
    const double ru = o2.x - up, rv = o2.y - vp;
    const double e = sqrt(fma(ru, ru, rv * rv));  // NaN if either coordinate is missing, like np.linalg.norm(obs - pred)
    const bool ok = e == e;
    sum += ok ? e : 0.0;
    cnt += ok ? 1.0 : 0.0;
    ep[(size_t)p * Fpad] = e;
  };
  int p = 0;
  for (; p + PF <= N; p += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const double2 o2 = ring[j];
      ring[j] = op[(size_t)min(p + j + PF, N - 1) * Fpad];
      point(o2, p + j);
    }
  }
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (p + j < N) point(ring[j], p + j);
  if (f < F) {
    mean_cf[(size_t)c * F + f] = cnt > 0.0 ? sum / cnt : __builtin_nan("");  // np.nanmean of an all-NaN row is NaN
    full_cf[(size_t)c * F + f] = full;
  }
}

// ---------------------------------------------------------------- exact order statistic by radix select
// st: [0] prefix  [1] rank  [2] count of candidates  [3] result bits ; hist 256 x u32 behind it
struct SelState {
  unsigned long long prefix, rank, count, value;
  unsigned int hist[256];
};

// values: [groups][stride] doubles with the frame index = i % Fpad; mask (per frame) may be nullptr; group g selects the
// slice [g * per_group, (g + 1) * per_group) (per-camera medians) -- blockIdx.y = group, one SelState per group
// dual != 0: two states per group (2 g: the lower middle rank, 2 g + 1: the upper one) walk the same slice in the same passes
// skey != 0: the values may be negative -- they are compared through an order-preserving key (sign bit flipped for values >= +0, every bit
// for negative ones: unsigned order of the keys = numeric order of the values); the errors of the pre-filter are >= +0 and need none
__device__ __forceinline__ unsigned long long sel_key(unsigned long long k, int skey) {
  return skey ? ((k >> 63) ? ~k : (k | 0x8000000000000000ull)) : k;
}
__global__ __launch_bounds__(256) void k_sel_hist(const double* __restrict__ v, const unsigned char* __restrict__ fmask, size_t per_group, int Fpad, SelState* __restrict__ sts, int pass, int dual, int skey) {
  __shared__ unsigned int s_h[256];
  SelState* st = sts + blockIdx.y;
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long prefix = st->prefix;
  const int shift_hi = 64 - 8 * pass, shift = 56 - 8 * pass;
  const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(v) + (size_t)(dual ? blockIdx.y >> 1 : blockIdx.y) * per_group;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_group; i += (size_t)gridDim.x * 256) {
    const unsigned long long kraw = keys[i];
    const double d = __longlong_as_double((long long)kraw);
    const unsigned long long k = sel_key(kraw, skey);
    bool ok = d == d;
    if (fmask) ok = ok && fmask[i % (size_t)Fpad] != 0;
    if (ok && (pass == 0 || (k >> shift_hi) == prefix)) atomicAdd(&s_h[(unsigned)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  const unsigned int n = s_h[threadIdx.x];
  if (n) atomicAdd(&st->hist[threadIdx.x], n);
}

// ---------------------------------------------------------------- the pre-filter's frame selection, all of it on the device
// (mcba_prefilter: bundle_adjustment.py:266-285 in ONE host synchronisation).  What the host did between k_frame_err and the
// gather -- which frames are complete in two cameras (:266), the worst camera's mean error per frame (:279), 5 x nanmedian of the
// per-point errors of those frames (:281-282), the comparison (:285) -- runs here; the host gets one status byte per frame.
// The median is the exact order statistic by radix select on the bit patterns, as above, but in THREE passes over the 26 MB
// error array instead of eight, and without the host in between:
//   pass 0  histogram of the leading 12 bits (sign + exponent; the lanes of a wavefront that share a digit -- nearly all of them --
//           post ONE LDS atomic) + the per-frame masks (which the later passes and the status kernel read);
//   pass 1  every workgroup picks the exponent bin of the two middle ranks from pass 0's histogram, then histograms the next 10 bits
//           of the values in it -- its OWN histogram, stored, not added, to a slot of its own; k_pf_sum adds the slots up;
//   pass 2  every workgroup picks the 22-bit prefix, then COMPACTS the values that carry it (LDS staging, one reservation per
//           workgroup in one of 8 candidate lists: ~2.5 k of 3.2 M values at 6 x 10 000 x 54);
//   final   one workgroup selects among the candidates (radix passes over the remaining 42 bits, the candidates in LDS) -> median;
//   status  exclusion per frame, packed behind the 8 info doubles for ONE device-to-host copy.
// What shaped it (rocprofv3, profiles/round5/NOTES_round5.md): (i) a first version with 2 198 small workgroups, a ticket per pass for
// "the last one picks" and one shared candidate counter ran 150-190 us PER PASS: device-scope atomics on ONE address retire at ~70 ns
// each on this part, whoever issues them; (ii) 128 workgroups adding 1 024-bin histograms into global replicas: 127 us -- ~1 ns per
// global atomic in aggregate; (iii) what a pass costs beyond that is memory round trips per wavefront (~2 us each), not bytes.  Hence:
// no tickets (the NEXT pass's workgroups redo the pick: kernel boundary = visibility), no global histogram atomics beyond the handful
// of exponent bins of pass 0 (8 replicas), candidates reserved once per workgroup in 8 lists, eight 512-byte loads in flight per
// wavefront.  A candidate list that overflows (all errors equal ...) sets `overflow`; the host then falls back to the eight-pass
// select above on the same mask.
constexpr int PF_G = 256;            // workgroups of the passes at most
constexpr int PF_REP = 8;            // replicas of pass 0's histogram / candidate lists
constexpr int PF_SEG = 1024;         // candidates a workgroup stages per order statistic
constexpr int PF_LIST = 32768;       // capacity of one candidate list
constexpr int PF_BITS_B = 10;        // digit of pass 1
constexpr int PF_NB = 1 << PF_BITS_B;
constexpr int PF_ROWS = 8;           // rows a wavefront takes of a work item (loads in flight)
constexpr int PF_NA = 2048;          // bins of pass 0: the 11 exponent bits (the errors are >= +0: the sign bit is clear)
struct PrefState {
  unsigned long long histA[PF_REP][PF_NA];
  unsigned long long histB[2][PF_NB];               // pass 1's histograms, summed over the workgroups' slots by k_pf_sum
  unsigned int list_count[2][PF_REP];
  unsigned int overflow, pad[3];
  unsigned long long total, rankA[2], prefixA[2];   // written by workgroup 0 of pass 1: the 12-bit bins of the two middle ranks, the ranks inside them
  unsigned long long rankB[2], prefixB[2];          // written by workgroup 0 of pass 2: the 20-bit prefixes, the ranks inside them
  double info[8];                                   // 0 threshold  1 median  2 number of values  3 overflow
  // (not cleared: every slot that is read has been written by the same call)
  unsigned int partB[PF_G][2][PF_NB];
  unsigned long long cand[2][PF_REP][PF_LIST];
};
size_t prefilter_state_bytes() { return sizeof(PrefState); }
size_t prefilter_state_clear_bytes() { return offsetof(PrefState, partB); }

// one wavefront: the bin that holds rank r of a histogram of NB bins (a multiple of 64; LDS or global), and the number of values below it
template <int NB, class T>
__device__ __forceinline__ void wave_pick_bin(const T* hist, unsigned long long r, int lane, unsigned& digit, unsigned long long& below) {
  constexpr int PER = NB / 64;
  unsigned long long cnt[PER];
  unsigned long long mine = 0;
#pragma unroll
  for (int b = 0; b < PER; ++b) { cnt[b] = (unsigned long long)hist[lane * PER + b]; mine += cnt[b]; }
  unsigned long long incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  const unsigned long long excl = incl - mine;
  const unsigned long long who = __ballot(r >= excl && r < incl);
  const int src = who ? __ffsll((long long)who) - 1 : 63;  // (empty histogram: the last lane -- never used: the callers check the total)
  unsigned d = lane * PER + PER - 1;
  unsigned long long bl = excl;
  bool found = false;
#pragma unroll
  for (int b = 0; b < PER; ++b) {
    if (!found) {
      if (r < bl + cnt[b] || b == PER - 1) { d = (unsigned)(lane * PER + b); found = true; }
      else bl += cnt[b];
    }
  }
  digit = __shfl(d, src, 64);
  below = __shfl(bl, src, 64);
}

// one WAVEFRONT per state: which byte holds the wanted rank (wave_pick_bin: four bins per lane, a prefix scan across the lanes); upper == 0
// selects rank (n-1)/2, upper == 1 rank n/2.  (Round 6: one thread walking the 256 bins with dependent global loads took 15 us per launch,
// eight launches per median -- 120 us of the 216 us pairwise-median crossing of calibrate().)
__global__ __launch_bounds__(64) void k_sel_pick(SelState* __restrict__ sts, int pass, int upper, int dual, int skey) {
  SelState* st = sts + blockIdx.x;
  const int lane = threadIdx.x;
  if (dual) upper = blockIdx.x & 1;
  unsigned long long mine = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) mine += st->hist[4 * lane + b];
  unsigned long long total = mine;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
  unsigned long long r = pass == 0 ? (total ? (upper ? total / 2 : (total - 1) / 2) : 0) : st->rank;
  unsigned digit;
  unsigned long long below;
  wave_pick_bin<256>(st->hist, r, lane, digit, below);
  if (total == 0 || r >= total) { digit = 0; below = 0; }   // (an empty group: the value is never used -- count 0 makes the caller report NaN)
  const unsigned long long prefix = ((pass == 0 ? 0ull : st->prefix) << 8) | (unsigned long long)digit;
  __syncthreads();   // (every lane has read its bins and the old state)
#pragma unroll
  for (int b = 0; b < 4; ++b) st->hist[4 * lane + b] = 0;
  if (lane == 0) {
    if (pass == 0) st->count = total;
    st->rank = r - below;
    st->prefix = prefix;
    if (pass == 7) st->value = skey ? ((prefix >> 63) ? (prefix & 0x7FFFFFFFFFFFFFFFull) : ~prefix) : prefix;   // (the key back to the value's bits)
  }
}

// per frame: used (complete in >= 2 cameras, :266), complete in every camera, the worst camera's nan-mean error (np.nanmax, :279)
__device__ __forceinline__ unsigned char pf_frame_mask(const double* __restrict__ mean_cf, const double* __restrict__ full_cf, int f, int C, int F, int N, double& worst) {
  int ncomp = 0;
  double w = __builtin_nan("");
  for (int c = 0; c < C; ++c) {
    ncomp += full_cf[(size_t)c * F + f] == (double)N ? 1 : 0;
    w = fmax(w, mean_cf[(size_t)c * F + f]);  // fmax ignores a NaN operand: NaN only where every camera is NaN (np.fmax.reduce)
  }
  worst = w;
  return (unsigned char)((ncomp > 1 ? 1 : 0) | (ncomp == C ? 4 : 0));
}
__global__ __launch_bounds__(256) void k_pf_mask(const double* __restrict__ mean_cf, const double* __restrict__ full_cf, unsigned char* __restrict__ fmask, unsigned char* __restrict__ status,
                                                 double* __restrict__ worst, int C, int F, int N, int Fpad) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= Fpad) return;
  if (f >= F) { fmask[f] = 0; return; }
  double w;
  const unsigned char s = pf_frame_mask(mean_cf, full_cf, f, C, F, N, w);
  fmask[f] = s & 1;
  status[f] = s;
  worst[f] = w;
}

// err: [R rows][Fpad].  Work items = (64-frame block, chunk of 16 x PF_ROWS rows); the 16 wavefronts of a workgroup take PF_ROWS rows of
// an item each; workgroup g takes the items g, g + G, ...
template <int PASS>
__global__ __launch_bounds__(1024) void k_pf_pass(const unsigned long long* __restrict__ keys, const double* __restrict__ mean_cf, const double* __restrict__ full_cf, unsigned char* __restrict__ fmask,
                                                  unsigned char* __restrict__ status, double* __restrict__ worst, int C, int F, int N, int Fpad, PrefState* __restrict__ st) {
  constexpr int NH = PASS == 0 ? PF_NA : (PASS == 1 ? 2 * PF_NB : 4 * PF_SEG);   // LDS words: histogram(s), or the two staged candidate lists (u64 = two words each)
  __shared__ __align__(8) unsigned int s_h[NH];   // (pass 2 stages its candidates here as 64-bit words)
  __shared__ unsigned long long s_sum[PASS == 1 ? PF_NA : 64];   // pass 1: pass 0's histogram, summed over its replicas
  __shared__ unsigned long long s_pre[2], s_rank[2];
  __shared__ unsigned int s_n[2], s_base[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int CH = 16 * PF_ROWS;
  const int R = C * N, nfb = Fpad / 64, nrc = (R + CH - 1) / CH, items = nfb * nrc;
  if (PASS <= 1) for (int b = threadIdx.x; b < NH; b += 1024) s_h[b] = 0;
  if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
  // ---- the previous pass's pick, by every workgroup (its histogram is complete: kernel boundary)
  if (PASS == 1) {
    for (int b = threadIdx.x; b < PF_NA; b += 1024) {
      unsigned long long c = 0;
#pragma unroll
      for (int k = 0; k < PF_REP; ++k) c += st->histA[k][b];
      s_sum[b] = c;
    }
    __syncthreads();
    if (wave == 0) {
      unsigned long long total = 0;
      for (int b = lane; b < PF_NA; b += 64) total += s_sum[b];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
      for (int s = 0; s < 2; ++s) {
        const unsigned long long r = total ? (s ? total / 2 : (total - 1) / 2) : 0;
        unsigned dg; unsigned long long below;
        wave_pick_bin<PF_NA>(s_sum, r, lane, dg, below);
        if (lane == 0) {
          s_pre[s] = dg; s_rank[s] = r - below;
          if (blockIdx.x == 0) { st->prefixA[s] = dg; st->rankA[s] = r - below; st->total = total; }
        }
      }
    }
  }
  if (PASS == 2) {   // pass 1's histograms were summed by k_pf_sum: pick the 22-bit prefixes
    const bool same = st->prefixA[1] == st->prefixA[0];
    if (wave < 2) {
      const int s = wave;
      unsigned dg; unsigned long long below;
      const unsigned long long r = st->rankA[s];
      wave_pick_bin<PF_NB>(&st->histB[same ? 0 : s][0], r, lane, dg, below);
      if (lane == 0) {
        const unsigned long long pre = (st->prefixA[s] << PF_BITS_B) | dg;
        s_pre[s] = pre; s_rank[s] = r - below;
        if (blockIdx.x == 0) { st->prefixB[s] = pre; st->rankB[s] = r - below; }
      }
    }
  }
  __syncthreads();
  const unsigned long long pre0 = PASS ? s_pre[0] : 0, pre1 = PASS ? s_pre[1] : 0;
  unsigned long long* s_c = reinterpret_cast<unsigned long long*>(s_h);   // PASS 2: [2][PF_SEG] staged candidates
  for (int it = blockIdx.x; it < items; it += gridDim.x) {
    const int fb = it % nfb, rc = it / nfb;
    const int f = fb * 64 + lane;
    const int r0 = rc * CH + wave * PF_ROWS;
    unsigned long long k8[PF_ROWS];
#pragma unroll
    for (int j = 0; j < PF_ROWS; ++j) k8[j] = r0 + j < R ? keys[(size_t)(r0 + j) * Fpad + f] : 0x7FF8000000000000ull;  // (past the end: NaN = not a value)
    bool m;
    if (PASS == 0) {
      double w = 0.0;
      const unsigned char sf = f < F ? pf_frame_mask(mean_cf, full_cf, f, C, F, N, w) : (unsigned char)0;
      m = (sf & 1) != 0;
      if (rc == 0 && wave == 0) {   // one wavefront per frame block publishes the masks for the later passes / the status kernel
        fmask[f] = sf & 1;
        if (f < F) { status[f] = sf; worst[f] = w; }
      }
    } else m = fmask[f] != 0;
#pragma unroll
    for (int j = 0; j < PF_ROWS; ++j) {
      const unsigned long long k = k8[j];
      const double d = __longlong_as_double((long long)k);
      const bool ok = m && d == d;
      if (PASS == 0) {
        const unsigned dig = (unsigned)(k >> 52) & (PF_NA - 1);
        unsigned long long rem = __ballot(ok);
        while (rem) {   // (errors of one order of magnitude share the exponent: without this every instruction is a 64-way conflict on one LDS word)
          const int lead = __ffsll((long long)rem) - 1;
          const unsigned dl = __shfl(dig, lead, 64);
          const unsigned long long same = __ballot(ok && dig == dl) & rem;
          if (lane == lead) atomicAdd(&s_h[dl], (unsigned)__popcll(same));
          rem &= ~same;
        }
      } else if (PASS == 1) {
        const unsigned hi = (unsigned)(k >> 52), dig = (unsigned)(k >> (52 - PF_BITS_B)) & (PF_NB - 1);
        if (ok && hi == (unsigned)pre0) atomicAdd(&s_h[dig], 1u);
        if (ok && hi == (unsigned)pre1 && pre1 != pre0) atomicAdd(&s_h[PF_NB + dig], 1u);   // (both ranks in one bin, the usual case: state 1 picks from state 0's histogram)
      } else {
        const unsigned long long p20 = k >> (52 - PF_BITS_B);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s == 1 && pre1 == pre0) break;   // (the usual case: both middle ranks in one bin -- one list serves both)
          const bool hit = ok && p20 == (s ? pre1 : pre0);
          const unsigned long long hm = __ballot(hit);
          if (hm) {
            unsigned base = 0;
            const int lead = __ffsll((long long)hm) - 1;
            if (lane == lead) base = atomicAdd(&s_n[s], (unsigned)__popcll(hm));
            base = __shfl(base, lead, 64);
            const unsigned idx = base + (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
            if (hit && idx < (unsigned)PF_SEG) s_c[s * PF_SEG + idx] = k;
          }
        }
      }
    }
  }
  __syncthreads();
  if (PASS == 0) {
    unsigned long long* dst = st->histA[blockIdx.x % PF_REP];
    for (int b = threadIdx.x; b < PF_NA; b += 1024) if (s_h[b]) atomicAdd(&dst[b], (unsigned long long)s_h[b]);   // (a handful of exponent bins per workgroup)
  } else if (PASS == 1) {
    for (int b = threadIdx.x; b < 2 * PF_NB; b += 1024) (&st->partB[blockIdx.x][0][0])[b] = s_h[b];
  } else {
    const int rep = blockIdx.x % PF_REP;
    if (threadIdx.x < 2) {
      const int s = threadIdx.x;
      const unsigned n = min(s_n[s], (unsigned)PF_SEG);
      unsigned base = n ? atomicAdd(&st->list_count[s][rep], n) : 0u;
      if (s_n[s] > (unsigned)PF_SEG || base + n > (unsigned)PF_LIST) { st->overflow = 1u; if (base + n > (unsigned)PF_LIST) base = PF_LIST; }   // (nothing is written past a full list)
      s_base[s] = base;
    }
    __syncthreads();
    for (int s = 0; s < 2; ++s) {
      const unsigned n = min(s_n[s], (unsigned)PF_SEG), base = s_base[s];
      for (unsigned i = threadIdx.x; i < n && base + i < (unsigned)PF_LIST; i += 1024) st->cand[s][rep][base + i] = s_c[s * PF_SEG + i];
    }
  }
}

// pass 1's per-workgroup histograms (slots) -> histB: 2 PF_NB / 256 workgroups, thread = (bin, quarter of the slots), all loads of a
// thread independent.  (Summing the slots in pass 2's prologue instead -- every workgroup for itself -- made that prologue 20 us.)
__global__ __launch_bounds__(1024) void k_pf_sum(PrefState* __restrict__ st, int nslots) {
  __shared__ unsigned long long s_p[4][256];
  const int bin = blockIdx.x * 256 + (threadIdx.x & 255), q = threadIdx.x >> 8;
  const unsigned int* src = &st->partB[0][0][0] + bin;
  unsigned long long c = 0;
  int g = q;
  for (; g + 28 < nslots; g += 32) {
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(g + 4 * j) * (2 * PF_NB)];
#pragma unroll
    for (int j = 0; j < 8; ++j) c += v[j];
  }
  for (; g < nslots; g += 4) c += src[(size_t)g * (2 * PF_NB)];
  s_p[q][threadIdx.x & 255] = c;
  __syncthreads();
  if (q == 0) (&st->histB[0][0])[bin] = s_p[0][threadIdx.x] + s_p[1][threadIdx.x] + s_p[2][threadIdx.x] + s_p[3][threadIdx.x];
}

// final selection among the candidates of each order statistic (one workgroup): radix passes of 11 bits over the bits below the
// prefix, both order statistics in the same sweep; the candidates of the 8 lists are gathered into LDS when they fit (dynamic LDS)
constexpr int PF_LDS_LIST = 16384;
size_t prefilter_final_lds_bytes() { return (size_t)PF_LDS_LIST * sizeof(unsigned long long); }
__global__ __launch_bounds__(1024) void k_pf_final(PrefState* __restrict__ st, double thr_scale, unsigned lds_capacity) {   // lds_capacity: candidates the dynamic LDS holds (0: none was granted -- the lists are read where they lie)
  extern __shared__ __align__(16) unsigned long long s_list[];
  __shared__ unsigned int s_h[2][2048];
  __shared__ unsigned int s_off[2][PF_REP + 1];
  __shared__ unsigned long long s_low[2], s_r[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LOWBITS = 52 - PF_BITS_B;
  const bool one_list = st->prefixB[0] == st->prefixB[1];
  if (threadIdx.x < 2) {   // offsets of the lists in the gathered one
    const int s = threadIdx.x;
    unsigned cnt[PF_REP];
#pragma unroll
    for (int k = 0; k < PF_REP; ++k) cnt[k] = min(st->list_count[s][k], (unsigned)PF_LIST);
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < PF_REP; ++k) { s_off[s][k] = acc; acc += cnt[k]; }
    s_off[s][PF_REP] = acc;
    s_low[s] = 0;
    s_r[s] = st->rankB[s];
  }
  __syncthreads();
  const unsigned n0 = s_off[0][PF_REP], n1 = one_list ? 0u : s_off[1][PF_REP];
  const bool in_lds = n0 + n1 <= lds_capacity;
  if (in_lds) {
    for (int k = 0; k < PF_REP; ++k) {
      const unsigned c0 = s_off[0][k + 1] - s_off[0][k];
      for (unsigned j = threadIdx.x; j < c0; j += 1024) s_list[s_off[0][k] + j] = st->cand[0][k][j];
      if (!one_list) {
        const unsigned c1 = s_off[1][k + 1] - s_off[1][k];
        for (unsigned j = threadIdx.x; j < c1; j += 1024) s_list[n0 + s_off[1][k] + j] = st->cand[1][k][j];
      }
    }
  }
  for (int remaining = LOWBITS; remaining > 0;) {
    const int nb = min(11, remaining), shift = remaining - nb;
    __syncthreads();
    for (int b = threadIdx.x; b < 4096; b += 1024) (&s_h[0][0])[b] = 0;
    __syncthreads();
    const unsigned long long low0 = s_low[0], low1 = s_low[1];
    auto count = [&](unsigned long long k, int list) {   // a candidate of `list` (0: of order statistic 0 -- and of 1 when one_list; 1: of order statistic 1)
      const unsigned long long lw = k & ((1ull << LOWBITS) - 1ull);
      const unsigned dg = (unsigned)(lw >> shift) & ((1u << nb) - 1u);
      const unsigned long long up = remaining == LOWBITS ? 0 : lw >> remaining;
      if (list == 0 && (remaining == LOWBITS || up == low0)) atomicAdd(&s_h[0][dg], 1u);
      if ((list == 1 || one_list) && (remaining == LOWBITS || up == low1)) atomicAdd(&s_h[1][dg], 1u);
    };
    if (in_lds) {
      for (unsigned i = threadIdx.x; i < n0 + n1; i += 1024) count(s_list[i], i < n0 ? 0 : 1);
    } else {
      for (int k = 0; k < PF_REP; ++k) {
        const unsigned c0 = s_off[0][k + 1] - s_off[0][k];
        for (unsigned j = threadIdx.x; j < c0; j += 1024) count(st->cand[0][k][j], 0);
        if (!one_list) {
          const unsigned c1 = s_off[1][k + 1] - s_off[1][k];
          for (unsigned j = threadIdx.x; j < c1; j += 1024) count(st->cand[1][k][j], 1);
        }
      }
    }
    __syncthreads();
    if (wave < 2) {
      unsigned dg; unsigned long long below;
      wave_pick_bin<2048>(&s_h[wave][0], s_r[wave], lane, dg, below);
      if (lane == 0) { s_r[wave] -= below; s_low[wave] = (s_low[wave] << nb) | dg; }
    }
    remaining = shift;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long total = st->total;
    const double a = __longlong_as_double((long long)((st->prefixB[0] << LOWBITS) | s_low[0])), b = __longlong_as_double((long long)((st->prefixB[1] << LOWBITS) | s_low[1]));
    const double med = total ? 0.5 * (a + b) : __builtin_nan("");   // np.nanmedian: mean of the two middle values
    st->info[0] = thr_scale * med;
    st->info[1] = med;
    st->info[2] = (double)total;
    st->info[3] = st->overflow ? 1.0 : 0.0;
  }
}

// exclusion (:285): np.nan_to_num(worst mean error) > threshold, for the used frames -- written straight into the packed result
// [info 8 doubles | status F bytes] that goes to the host in one copy
__global__ __launch_bounds__(256) void k_pf_status(const double* __restrict__ worst, unsigned char* __restrict__ status, unsigned char* __restrict__ packed, int F, PrefState* __restrict__ st, double thr_arg,
                                                   int thr_from_state) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  const double thr = thr_from_state ? st->info[0] : thr_arg;
  if (f < F) {
    unsigned char s = status[f] & 5;
    double w = worst[f];
    w = w != w ? 0.0 : (isinf(w) ? (w > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308) : w);   // np.nan_to_num
    if ((s & 1) && w > thr) s |= 2;
    status[f] = s;
    packed[64 + f] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) {
    double v = st->info[threadIdx.x];
    if (threadIdx.x == 0 && !thr_from_state) v = thr_arg;
    reinterpret_cast<double*>(packed)[threadIdx.x] = v;
  }
}

// parameter vector of a frame subset: the camera blocks, then the poses of the chosen frames
__global__ __launch_bounds__(256) void k_gather_params(const double* __restrict__ xs, const int* __restrict__ frames, double* __restrict__ xd, int C, int Fdst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ncam = 12 * C;
  if (i < ncam) xd[i] = xs[i];
  else if (i < ncam + 6 * Fdst) {
    const int j = i - ncam, f = j / 6, k = j - 6 * f;
    xd[i] = xs[ncam + 6 * (size_t)frames[f] + k];
  }
}

// ---------------------------------------------------------------- frame subsets, device to device
__global__ void k_gather_frames(const double2* __restrict__ src, const int* __restrict__ frames, double2* __restrict__ dst, int C, int Fsrc, int Fdst, int N, const int* __restrict__ only_cam) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)C * Fdst * N;
  if (i >= total) return;
  const int p = (int)(i % N);
  const size_t cf = i / N;
  const int f = (int)(cf % Fdst), c = (int)(cf / Fdst);
  const double nan = __builtin_nan("");
  dst[i] = (only_cam && only_cam[f] != c) ? double2{nan, nan} : src[((size_t)c * Fsrc + frames[f]) * N + p];
}

// ---------------------------------------------------------------- presence bits of the observation scalars
// bit i (numpy.packbits order: scalar 8 b + j is bit 7 - j of byte b) = scalar i of the (camera, frame, point, u|v) array is not NaN.
// One wavefront = 64 consecutive scalars = one 8-byte word: the ballot, bit-reversed and byte-swapped.
__global__ __launch_bounds__(256) void k_seen_bits(const double* __restrict__ obs_raw, size_t count, unsigned long long* __restrict__ words) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double v = i < count ? __builtin_nontemporal_load(obs_raw + i) : __builtin_nan("");
  const unsigned long long b = __ballot(v == v);
  if ((threadIdx.x & 63) == 0 && i < count) words[i >> 6] = __builtin_bswap64(__brevll(b));  // (the last block's wavefronts past the end own no word)
}

// ---------------------------------------------------------------- undistortion (cv2.undistortPoints(src, K, dist, None, K))
__device__ __forceinline__ void undistort_px(double u, double v, double fx, double fy, double cx, double cy, const double* k, int iters, double& uo, double& vo) {
  const double x0 = (u - cx) / fx, y0 = (v - cy) / fy;
  double x = x0, y = y0;
  for (int it = 0; it < iters; ++it) {
    const double r2 = fma(x, x, y * y);
    const double icdist = 1.0 / fma(fma(fma(k[4], r2, k[1]), r2, k[0]), r2, 1.0);
    const double dx = fma(2.0 * k[2] * x, y, k[3] * fma(2.0 * x, x, r2));
    const double dy = fma(k[2], fma(2.0 * y, y, r2), 2.0 * k[3] * x * y);
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  uo = fma(x, fx, cx);
  vo = fma(y, fy, cy);
}

struct UndistCam {
  double K[4];
  double dist[5];
};

__global__ __launch_bounds__(256) void k_undistort(const double2* __restrict__ uv, double2* __restrict__ out, size_t n, UndistCam cam, int iters) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 o = uv[i];
  double2 r;
  undistort_px(o.x, o.y, cam.K[0], cam.K[1], cam.K[2], cam.K[3], cam.dist, iters, r.x, r.y);
  const bool ok = o.x == o.x && o.y == o.y;  // a point with a missing coordinate stays NaN in both (geometry.py:351-352)
  r.x = ok ? r.x : __builtin_nan("");
  r.y = ok ? r.y : __builtin_nan("");
  out[i] = r;
}

// ---------------------------------------------------------------- reprojection diagnostics
// 8x8 symmetric positive definite solve in registers (upper triangle packed row-major), in place on b
__device__ __forceinline__ int tri8(int i, int j) { return i * 8 - (i * (i - 1)) / 2 + (j - i); }
__device__ __forceinline__ bool chol_solve8(double* A, double* b) {
  bool ok = true;
  // A = L L^T, L stored over the upper triangle as L^T (row i = column i of L)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = i; j < 8; ++j) {
      double s = A[tri8(i, j)];
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(-A[tri8(k, i)], A[tri8(k, j)], s);
      if (j == i) {
        ok = ok && s > 0.0;
        A[tri8(i, i)] = sqrt(s > 0.0 ? s : 1.0);
      } else {
        A[tri8(i, j)] = s / A[tri8(i, i)];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = fma(-A[tri8(k, i)], b[k], s);
    b[i] = s / A[tri8(i, i)];
  }
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int k = i + 1; k < 8; ++k) s = fma(-A[tri8(i, k)], b[k], s);
    b[i] = s / A[tri8(i, i)];
  }
  return ok;
}

struct DiagCams {   // distortion of every camera (the solver's parameter vector carries k1, k2 only)
  double dist[40][5];
};

// lane = (camera c, frame f).  obs_t [C][N][Fpad]; x = parameter vector; board = objpoints (N,3), bn = {mean x, mean y, scale}
// of the board's XY (Hartley normalisation, computed once on the host); und [C][N][Fpad] (u,v) scratch for the undistorted
// detections.  Outputs (C,F,N,2): repro (distortion-free projection), trans (reprojection mapped to the board plane; NaN for
// (camera, frame) pairs with an incomplete detection), and err [C][N][Fpad] = |trans - board| (NaN likewise) for the
// per-camera medians.
// Homography: normalised inhomogeneous DLT (h33 = 1; 8x8 normal equations) as the start, then Levenberg-Marquardt on the
// transfer error in the board plane -- the quantity OpenCV's findHomography refines -- with a FIXED number of rounds (every
// lane runs the same instruction stream; a rejected step only raises that lane's damping).
__global__ __launch_bounds__(256) void k_reproj_diag(const double2* __restrict__ obs_t, const double* __restrict__ obj, const double* __restrict__ x, DiagCams dc, const double* __restrict__ bn,
                                                     double2* __restrict__ und, double* __restrict__ repro, double* __restrict__ trans, double* __restrict__ err, int C, int F, int N, int Fpad, int nfb, int iters,
                                                     int lm_iters) {
  __shared__ CamConst s_cam;
  const int c = blockIdx.y;
  if (threadIdx.x == 0) make_cam_const(x + 12 * c, s_cam);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fb = blockIdx.x * 4 + wave;
  if (fb >= nfb) return;
  const int f = fb * 64 + lane;
  const double fx = diag_uni(s_cam.fx), fy = diag_uni(s_cam.fy), cx = diag_uni(s_cam.cx), cy = diag_uni(s_cam.cy);
  double Rc[9], tc[3], kd[5];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rc[i] = diag_uni(s_cam.R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) tc[i] = diag_uni(s_cam.t[i]);
#pragma unroll
  for (int i = 0; i < 5; ++i) kd[i] = dc.dist[c][i];
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
  const double2* op = obs_t + (size_t)c * N * Fpad + f;
  double2* up = und + (size_t)c * N * Fpad + f;
  const double bmx = bn[0], bmy = bn[1], bs = bn[2];

  // pass 1: undistort; is the detection complete (viz.py:171: all 2N scalars present)?  centroid, then scale (Hartley)
  bool complete = f < F;
  double mx = 0.0, my = 0.0;
  for (int p = 0; p < N; ++p) {
    const double2 o = op[(size_t)p * Fpad];
    complete = complete && o.x == o.x && o.y == o.y;
    double2 u;
    undistort_px(o.x, o.y, fx, fy, cx, cy, kd, iters, u.x, u.y);
    up[(size_t)p * Fpad] = u;
    mx += u.x; my += u.y;
  }
  mx /= N; my /= N;
  double md = 0.0;
  for (int p = 0; p < N; ++p) {
    const double2 u = up[(size_t)p * Fpad];
    md += sqrt(fma(u.x - mx, u.x - mx, (u.y - my) * (u.y - my)));
  }
  const double ss = complete ? sqrt(2.0) * N / md : 1.0;
  // point p in normalised coordinates; incomplete lanes work on (0, 0): finite arithmetic, results discarded
  auto src = [&](int p, double& sx, double& sy, double& X, double& Y) {
    const double2 u = up[(size_t)p * Fpad];
    sx = complete ? (u.x - mx) * ss : 0.0;
    sy = complete ? (u.y - my) * ss : 0.0;
    X = (obj[3 * p] - bmx) * bs;
    Y = (obj[3 * p + 1] - bmy) * bs;
  };
  double h[8];
  {  // start: rows [s 1 0 0 0 -X s] h = X, [0 0 0 s 1 -Y s] h = Y  (s = (sx, sy))
    double A[36], b[8];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = 0.0;
    for (int p = 0; p < N; ++p) {
      double sx, sy, X, Y;
      src(p, sx, sy, X, Y);
      const double r0[8] = {sx, sy, 1.0, 0.0, 0.0, 0.0, -X * sx, -X * sy};
      const double r1[8] = {0.0, 0.0, 0.0, sx, sy, 1.0, -Y * sx, -Y * sy};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = i; j < 8; ++j) A[tri8(i, j)] = fma(r0[i], r0[j], fma(r1[i], r1[j], A[tri8(i, j)]));
        b[i] = fma(r0[i], X, fma(r1[i], Y, b[i]));
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) A[tri8(i, i)] = complete ? A[tri8(i, i)] : 1.0;
    const bool ok = chol_solve8(A, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = ok ? b[i] : ((i == 0 || i == 4) ? 1.0 : 0.0);  // degenerate detection: start from the identity
  }
  auto transfer_error = [&](const double* hh) {
    double e = 0.0;
    for (int p = 0; p < N; ++p) {
      double sx, sy, X, Y;
      src(p, sx, sy, X, Y);
      const double iw = 1.0 / fma(hh[6], sx, fma(hh[7], sy, 1.0));
      const double ex = X - fma(hh[0], sx, fma(hh[1], sy, hh[2])) * iw, ey = Y - fma(hh[3], sx, fma(hh[4], sy, hh[5])) * iw;
      e = fma(ex, ex, fma(ey, ey, e));
    }
    return e;
  };
  double e_cur = transfer_error(h), mu = 1e-4;
  for (int it = 0; it < lm_iters; ++it) {
    double A[36], g[8];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = 0.0;
    for (int p = 0; p < N; ++p) {
      double sx, sy, X, Y;
      src(p, sx, sy, X, Y);
      const double iw = 1.0 / fma(h[6], sx, fma(h[7], sy, 1.0));
      const double px = fma(h[0], sx, fma(h[1], sy, h[2])) * iw, py = fma(h[3], sx, fma(h[4], sy, h[5])) * iw;
      const double r0[8] = {sx * iw, sy * iw, iw, 0.0, 0.0, 0.0, -px * sx * iw, -px * sy * iw};
      const double r1[8] = {0.0, 0.0, 0.0, sx * iw, sy * iw, iw, -py * sx * iw, -py * sy * iw};
      const double ex = X - px, ey = Y - py;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = i; j < 8; ++j) A[tri8(i, j)] = fma(r0[i], r0[j], fma(r1[i], r1[j], A[tri8(i, j)]));
        g[i] = fma(r0[i], ex, fma(r1[i], ey, g[i]));
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) A[tri8(i, i)] = complete ? A[tri8(i, i)] * (1.0 + mu) : 1.0;   // Marquardt damping
    const bool ok = chol_solve8(A, g);
    double hn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hn[i] = h[i] + (ok ? g[i] : 0.0);
    const double e_new = transfer_error(hn);
    const bool accept = ok && e_new <= e_cur;   // (NaN compares false)
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = accept ? hn[i] : h[i];
    e_cur = accept ? e_new : e_cur;
    mu = accept ? fmax(mu * 0.1, 1e-15) : fmin(mu * 10.0, 1e8);
  }
  // last pass: distortion-free projection of the board (viz.py:166-168) mapped through the homography, distance to the board
  const double nan = __builtin_nan("");
  double* ep = err + (size_t)c * N * Fpad + f;
  for (int p = 0; p < N; ++p) {
    const double Xo[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
    const double xc = fma(pc.Rcf[0], Xo[0], fma(pc.Rcf[1], Xo[1], fma(pc.Rcf[2], Xo[2], pc.tcf[0])));
    const double yc = fma(pc.Rcf[3], Xo[0], fma(pc.Rcf[4], Xo[1], fma(pc.Rcf[5], Xo[2], pc.tcf[1])));
    const double zc = fma(pc.Rcf[6], Xo[0], fma(pc.Rcf[7], Xo[1], fma(pc.Rcf[8], Xo[2], pc.tcf[2])));
    const double ru = fma(fx, xc / zc, cx), rv = fma(fy, yc / zc, cy);
    const double sx = (ru - mx) * ss, sy = (rv - my) * ss;
    const double iw = 1.0 / fma(h[6], sx, fma(h[7], sy, 1.0));
    const double tx = fma(h[0], sx, fma(h[1], sy, h[2])) * iw / bs + bmx, ty = fma(h[3], sx, fma(h[4], sy, h[5])) * iw / bs + bmy;
    const double e = sqrt(fma(tx - Xo[0], tx - Xo[0], (ty - Xo[1]) * (ty - Xo[1])));
    ep[(size_t)p * Fpad] = complete ? e : nan;
    if (f < F) {
      const size_t o = (((size_t)c * F + f) * N + p) * 2;
      if (repro) { repro[o] = ru; repro[o + 1] = rv; }
      if (trans) { trans[o] = complete ? tx : nan; trans[o + 1] = complete ? ty : nan; }
    }
  }
}

// ---------------------------------------------------------------- launch wrappers
void launch_frame_err(hipStream_t st, const double* obs_t, const double* obj, const double* x, double* err, double* mean_cf, double* full_cf, int C, int F, int N, int Fpad, void* prefilter_state) {
  const int nfb = Fpad / 64;
  static_assert(offsetof(PrefState, partB) % 8 == 0, "the cleared part of the selection's state is whole 64-bit words");
  k_frame_err<<<dim3((nfb + 3) / 4, C), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, err, mean_cf, full_cf, C, F, N, Fpad, nfb,
                                                            static_cast<unsigned long long*>(prefilter_state), (unsigned)(prefilter_state_clear_bytes() / 8));
}

size_t select_state_bytes(int groups) { return (size_t)groups * sizeof(SelState); }

// median(s) of `groups` equal slices of v (per_group doubles each, frame = index % Fpad): after the call sel[g].count and
// sel[g].value (bit pattern of the order statistic) are valid.  16 tiny launches per call, no host synchronisation.
// upper = 0 / 1: one order statistic per group; upper = 2: BOTH middle ranks in the same eight passes (states 2 g and 2 g + 1:
// `sel` must hold 2 x groups states) -- half the launches and one host synchronisation instead of two for a median
void launch_select(hipStream_t st, const double* v, const unsigned char* fmask, size_t per_group, int groups, int Fpad, void* sel, int upper, int skey) {
  SelState* s = static_cast<SelState*>(sel);
  const int dual = upper == 2 ? 1 : 0, nst = dual ? 2 * groups : groups;
  (void)hipMemsetAsync(s, 0, select_state_bytes(nst), st);
  const unsigned bx = (unsigned)std::min<size_t>((per_group + 255) / 256, 1024);
  for (int pass = 0; pass < 8; ++pass) {
    k_sel_hist<<<dim3(bx, nst), dim3(256), 0, st>>>(v, fmask, per_group, Fpad, s, pass, dual, skey);
    k_sel_pick<<<dim3(nst), dim3(64), 0, st>>>(s, pass, upper, dual, skey);
  }
}

// one histogram pass with a host-given prefix (sharded select: the caller combines the histograms of several handles)
int launch_select_hist(hipStream_t st, const double* v, const unsigned char* fmask, size_t per_group, int Fpad, void* sel, unsigned long long prefix, int pass, unsigned int* hist256) {
  SelState* s = static_cast<SelState*>(sel);
  SelState init;
  memset(&init, 0, sizeof(init));
  init.prefix = prefix;
  if (hipMemcpyAsync(s, &init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
  const unsigned bx = (unsigned)std::min<size_t>((per_group + 255) / 256, 1024);
  k_sel_hist<<<dim3(bx, 1), dim3(256), 0, st>>>(v, fmask, per_group, Fpad, s, pass, 0, 0);
  if (hipMemcpyAsync(hist256, s->hist, 256 * sizeof(unsigned int), hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
  return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------- measurement aid: the FP64 vector issue rate this GPU sustains
// One wavefront per SIMD on every CU (the occupancy of the fused k_gram), each issuing independent v_fma_f64 with three distinct register
// pairs per instruction -- the operand pattern of the Gram update acc[6 i + j] += u[i] * v[j] (scripts/micro/fma_f64_operands.hip, MODE 1).
// bench.py prices k_gram's FP64 instruction stream against what this returns, next to the datasheet peak.
__global__ __launch_bounds__(256) void k_fp64_issue_rate(double* out, int iters, double a, double b) {
  double acc[36], u[6], v[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 6; ++i) { u[i] = a + 1e-3 * (threadIdx.x + i); v[i] = b + 1e-4 * ((int)threadIdx.x - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[6 * i + j]) : "v"(u[i]), "v"(v[j]));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 36; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// -> TFLOP/s (2 flop per FMA) with `ncu` workgroups of four wavefronts; 0 on failure
double measure_fp64_issue_rate(int ncu) {
  double* out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&out), (size_t)ncu * 256 * sizeof(double)) != hipSuccess) return 0.0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 4000;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {  // (the first repetition also ramps the clocks)
    (void)hipEventRecord(e0, nullptr);
    k_fp64_issue_rate<<<dim3(ncu), dim3(256), 0, nullptr>>>(out, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess) { best = 0.f; break; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  if (!(best > 0.f) || best > 1e29f) return 0.0;
  return (double)ncu * 4 * 64 * (double)iters * 36 * 2 / (best * 1e-3) / 1e12;
}

// The pre-filter's selection behind k_frame_err (mean_cf / full_cf / err as it left them): fmask [Fpad], status [F] (bit 0 used, bit 1
// excluded as an outlier, bit 2 complete in every camera), worst [F]; `state` = prefilter_state_bytes() of scratch.  threshold NaN:
// 5 x nanmedian of the used frames' per-point errors (three passes over err), else the caller's.  The result [info 8 doubles | status]
// lands in `packed` (64 + F bytes) for one device-to-host copy.  No synchronisation.
void launch_prefilter_select(hipStream_t st, const double* err, const double* mean_cf, const double* full_cf, unsigned char* fmask, unsigned char* status, double* worst, void* state,
                             unsigned char* packed, int C, int F, int N, int Fpad, double threshold, bool state_cleared) {
  PrefState* ps = static_cast<PrefState*>(state);
  if (!state_cleared) (void)hipMemsetAsync(ps, 0, prefilter_state_clear_bytes(), st);   // (mcba_prefilter: launch_frame_err cleared it)
  const bool median = threshold != threshold;
  if (median) {
    const int R = C * N, nfb = Fpad / 64, items = nfb * ((R + 16 * PF_ROWS - 1) / (16 * PF_ROWS));
    const int g = std::min(PF_G, items);
    bool big_lds;
    {  // the dynamic-LDS limit of k_pf_final (128 KiB next to 16 KiB of static LDS), raised once per device; a part that does not grant it
       // runs the final selection on the candidate lists where they lie (same result: ADVICE r5)
      static std::mutex mu;
      static unsigned char lds_state[64] = {};   // 0 not asked yet, 1 granted, 2 refused
      int dev = 0;
      (void)hipGetDevice(&dev);
      std::lock_guard<std::mutex> lk(mu);
      unsigned char& stt = lds_state[dev & 63];
      if (!stt) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pf_final), hipFuncAttributeMaxDynamicSharedMemorySize, (int)prefilter_final_lds_bytes());
        if (e != hipSuccess) (void)hipGetLastError();
        stt = e == hipSuccess ? 1 : 2;
      }
      big_lds = stt == 1;
    }
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(err);
    k_pf_pass<0><<<dim3(g), dim3(1024), 0, st>>>(keys, mean_cf, full_cf, fmask, status, worst, C, F, N, Fpad, ps);
    k_pf_pass<1><<<dim3(g), dim3(1024), 0, st>>>(keys, mean_cf, full_cf, fmask, status, worst, C, F, N, Fpad, ps);
    k_pf_sum<<<dim3(2 * PF_NB / 256), dim3(1024), 0, st>>>(ps, g);
    k_pf_pass<2><<<dim3(g), dim3(1024), 0, st>>>(keys, mean_cf, full_cf, fmask, status, worst, C, F, N, Fpad, ps);
    if (big_lds) k_pf_final<<<dim3(1), dim3(1024), prefilter_final_lds_bytes(), st>>>(ps, 5.0, (unsigned)PF_LDS_LIST);
    else k_pf_final<<<dim3(1), dim3(1024), 0, st>>>(ps, 5.0, 0u);
  } else {
    k_pf_mask<<<dim3((Fpad + 255) / 256), dim3(256), 0, st>>>(mean_cf, full_cf, fmask, status, worst, C, F, N, Fpad);
  }
  launch_prefilter_status(st, status, worst, state, packed, F, threshold, median ? 1 : 0);
}
// (also the second half of the fallback: the eight-pass select found the median, the host passes 5 x that)
void launch_prefilter_status(hipStream_t st, unsigned char* status, const double* worst, void* state, unsigned char* packed, int F, double threshold, int from_state) {
  k_pf_status<<<dim3((F + 255) / 256), dim3(256), 0, st>>>(worst, status, packed, F, static_cast<PrefState*>(state), threshold, from_state);
}

// a small host array written to device memory THROUGH THE KERNEL-ARGUMENT SEGMENT (<= 480 doubles): no staging copy, no wait -- a
// hipMemcpyAsync from pageable memory blocks the caller for ~15-25 us whatever its size (scripts/micro/h2d_pipeline.hip)
struct SmallVec { double v[480]; };
__global__ void k_store_small(double* __restrict__ dst, SmallVec sv, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = sv.v[i];
}
bool launch_store_small(hipStream_t st, double* dst, const double* src_host, size_t n) {
  if (n > 480) return false;
  SmallVec sv;
  memcpy(sv.v, src_host, n * sizeof(double));
  k_store_small<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(dst, sv, (int)n);
  return true;
}

// box constraints (mcba_set_bounds): the trial point of a step, projected onto the box
__global__ __launch_bounds__(256) void k_clip(double* __restrict__ x, const double* __restrict__ lo, const double* __restrict__ hi, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = fmin(fmax(x[i], lo[i]), hi[i]);
}
void launch_clip(hipStream_t st, double* x, const double* lo, const double* hi, size_t n) {
  k_clip<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(x, lo, hi, n);
}

void launch_gather_params(hipStream_t st, const double* x_src, const int* frames, double* x_dst, int C, int Fdst) {
  const int n = 12 * C + 6 * Fdst;
  k_gather_params<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(x_src, frames, x_dst, C, Fdst);
}

void launch_gather_frames(hipStream_t st, const double* src_raw, const int* frames, double* dst_raw, int C, int Fsrc, int Fdst, int N, const int* only_cam) {
  const size_t total = (size_t)C * Fdst * N;
  k_gather_frames<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(src_raw), frames, reinterpret_cast<double2*>(dst_raw), C, Fsrc, Fdst, N, only_cam);
}

void launch_seen_bits(hipStream_t st, const double* obs_raw, size_t count, unsigned long long* words) {
  if (count) k_seen_bits<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st>>>(obs_raw, count, words);
}

void launch_undistort(hipStream_t st, const double* uv, double* out, size_t n, const double* K4, const double* dist5, int iters) {
  UndistCam cam;
  for (int i = 0; i < 4; ++i) cam.K[i] = K4[i];
  for (int i = 0; i < 5; ++i) cam.dist[i] = dist5 ? dist5[i] : 0.0;
  k_undistort<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(uv), reinterpret_cast<double2*>(out), n, cam, iters);
}

void launch_reproj_diag(hipStream_t st, const double* obs_t, const double* obj, const double* x, const double* dist5, const double* bn, double* und, double* repro, double* trans, double* err, int C, int F, int N,
                        int Fpad, int iters, int lm_iters) {
  DiagCams dc;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < 5; ++i) dc.dist[c][i] = dist5[5 * c + i];
  const int nfb = Fpad / 64;
  k_reproj_diag<<<dim3((nfb + 3) / 4, C), dim3(256), 0, st>>>(reinterpret_cast<const double2*>(obs_t), obj, x, dc, bn, reinterpret_cast<double2*>(und), repro, trans, err, C, F, N, Fpad, nfb, iters, lm_iters);
}

}  // namespace mcba
