// mcba_backsub.h -- body of k_backsub (frame steps + trial parameters), shared by the stand-alone kernel (mcba_kernels.hip) and
// by k_solve_backsub (mcba_solve.hip), where the same workgroups run NEXT to the reduced solve and wait for its result.
//
// lane = frame.  t = g_f + W_f^T d_c with W read from the wave tiles (each load = 64 consecutive frames, 1 KiB),
// d_f = -(L L^T)^-1 t with the Cholesky factor k_syrk left in fbuf, x_dst = x_src + d.
// Per-block partials of  sum d^T(lambda D d - g_f),  sum |d_f|^2,  sum |x_f|^2.
// DcSrc: the camera step either rides in the kernel-argument segment (CamStep, host solve) or sits in device memory
// where k_solve_cam left it (DevStep); both are wave-uniform loads.
// Workgroup = 64 frames x BW wavefronts: wavefront w accumulates W_cf^T d_c for the cameras c = w, w + BW, ... (each
// load = 64 consecutive frames, 1 KiB), the partial 6-vectors meet in LDS, wavefront 0 finishes the frame solve.
#pragma once
#include "mcba_device.h"
#include "mcba_math.h"

namespace mcba {

struct DevStep {
  const double* __restrict__ v;
};
constexpr int kBacksubWaves = 8;

// k_solve_backsub: the camera step does not exist yet when these workgroups start.  What does not depend on it -- the W
// blocks (39 MB at 6 x 10 000: the whole memory time of this kernel), the frame factors, the poses -- is requested first;
// which buffers are current is already final in `early` (the state k_syrk published after its decision; the solve does not
// touch the slot bit, and changes the damping only when it fails).  The solve releases ONE word twice, built on the tick's
// sequence number: as soon as the camera step is in memory, and at its very end, when the state is final (every exit of the
// solve posts the final value; exits without a step skip the first).  The workgroup computes the frame steps after the first
// and stores them after the second, if the final state still wants a trial step.  Polls are bounded, so a solve that never ran
// cannot hang the grid.
struct BacksubWait {
  const double* early;  // LM state after the tick's decision
  const double* flag;   // one word the solve releases: 4 seq + 1 camera step in memory, + 2 state final, + 3 state final and no step
  const double* dc;     // the camera step the solve writes before it releases the flag
  double* mail;         // LDS, 8 + 12 C doubles: what the polling wavefront fetched, for the others
  double seq;
  int max_polls;
  // frame-sharded ticks with ONE collective: `early` is the state BEFORE the decision (the solve of this launch takes it).  The
  // loads go to the buffers an ACCEPTED step with the predicted damping makes current -- the prediction the speculative Schur
  // reduction was built on; if it fails the solve marks the next tick rebuild-only and the steps computed here are dropped.
  int spec;
  double lam_min;
  // A poll that runs out (the solve never posted within max_polls) must not pass silently: the trial slot is then stale or half
  // written.  The workgroup stamps the tick's sequence number into a device word -- the NEXT tick's decision (k_syrk, or the
  // solve of a one-collective tick) discards its trial point and only rebuilds the system when it finds its predecessor's number
  // there -- and into a host-mapped word, on which mcba_lm_auto_wait switches the handle to the two-launch path.
  double* timeout_dev;
  double* timeout_host;
  double dec_floor;     // spec != 0: floor of Nielsen's factor the prediction assumes
  int strict;           // != 0 (the default): the reader acquires the release word with an agent-scope fence (release_word_acquired; MCBA_STRICT_SYNC=0 / mcba_set_strict_sync(h, 0) for the relaxed form)
};

__device__ __forceinline__ void backsub_stamp_timeout(const BacksubWait* w) {
  __hip_atomic_store(w->timeout_dev, w->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (w->timeout_host) __hip_atomic_store(w->timeout_host, w->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double load_coherent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- the READER SIDE of the release word, in one place.
// The solve releases the word with an agent-scope release store after a fence (mcba_solve.hip: post_state).  A reader polls it with
// release_word_poll and then calls release_word_acquired() ONCE before it touches anything the word guards.  What that does:
//   strict (DEFAULT    an agent-scope ACQUIRE fence (= an L2 invalidation) in the polling wavefront: the form the HIP memory model asks for --
//   since round 6)     the relaxed poll + this fence synchronise with the solve's release store, everything read afterwards is ordered behind
//                      it.  157 fences per launch at 6 x 10 000: +1.3 us per iteration (profiles/round5/bench_r5h.json: 102.5 -> 103.8 us;
//                      round 3 measured +2.5).  Round 6 made it what ships: 1.3 % of an iteration buys a product that is race-free by the
//                      model it is written in, and a driver test run that exercises what users run.
//   relaxed            selected AT RUN TIME per handle (MCBA_STRICT_SYNC=0 in the environment when the handle is created, or
//                      mcba_set_strict_sync(h, 0)): a COMPILER barrier only.  Everything read behind the word -- the camera step, DONE / SKIP
//                      of the LM state -- is fetched with load_coherent (agent-scope relaxed atomic loads: they bypass the per-XCD L2, which
//                      is not coherent across XCDs, so no stale line can be served and no invalidation is needed), the hardware issues a
//                      wavefront's loads in order, and only ONE wavefront per workgroup reads (the others get the values through LDS behind
//                      a barrier).  Under the HIP memory model this is a data race (relaxed loads do not synchronise with the release): it
//                      relies on gfx950 behaviour; tests/test_gpu_parity_large.py::test_fused_backsub_full_size_bit_identical stresses both
//                      forms at more than one waiting workgroup per CU (157 / 469 workgroups x 150 ticks, bit-identical iterates).
__device__ __forceinline__ void release_word_acquired(int strict) {
  if (strict) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else asm volatile("" ::: "memory");
}
// polls until the word equals one of `a`, `b` (returns 1), equals `stop` (returns 0) or max_polls ran out (returns -1)
__device__ __forceinline__ int release_word_poll(const double* word, double a, double b, double stop, int max_polls) {
  for (int polls = 0;; ) {
    const double v = load_coherent(word);
    if (v == a || v == b) return 1;
    if (v == stop) return 0;
    if (++polls > max_polls) return -1;
    __builtin_amdgcn_s_sleep(4);
  }
}

// CW: camera block width -- 12, or 6 = the intrinsics of every camera are held fixed: the camera step has 6 entries per camera
// (rho, t), only the W rows 6..11 of a record are read (18 of its 36 double2 rows), the intrinsics are copied to the trial slot.
template <class DcSrc, int CW = 12>
__device__ __forceinline__ void backsub_body(Sel sl, const double* __restrict__ rec0, const double* __restrict__ rec1, const double* __restrict__ fbuf, const DcSrc dcs, double* __restrict__ x0,
                                             double* __restrict__ x1, double* __restrict__ bpart, int C, int F, int Fpad, int block, int nw, double (*s_t)[6][64], const BacksubWait* wait) {
  const int n = 12 * C, nfb = Fpad >> 6;  // n: where the frame blocks start in x (the layout does not change with CW)
  constexpr int NR = 3 * CW, R0 = 3 * (12 - CW);  // double2 rows of a record's W block that are read: R0 .. R0 + NR
  const int nc = CW * C;                  // entries of the camera step
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nw) return;  // (k_solve_backsub launches more wavefronts than a small rig has cameras)
  const int f = block * 64 + lane;
  const bool fin = wave == 0 && f < F;
#define FSTAMP(k) do { } while (0)
  double2 v[NR];
  double xv[6], Lp[21], gf[6], D[6];
  unsigned frozen = 0;
  int sidx_early = 0;
  if (wait) {
    const bool flip = wait->spec && wait->early[MCBA_LM_SKIP] == 0.0;  // (a rebuild-only tick decides nothing)
    sidx_early = (static_cast<int>(wait->early[3]) ^ sl.idx ^ (flip ? 1 : 0)) & 1;
    if (wave < C) {
      const double2* w2 = reinterpret_cast<const double2*>((sidx_early ? rec1 : rec0) + ((size_t)wave * nfb + block) * (MCBA_REC * 64)) + lane;
#pragma unroll
      for (int k = 0; k < NR; ++k) v[k] = w2[(R0 + k) * 64];
    }
    if (fin) {
      const double* xf = (sidx_early ? x1 : x0) + n + 6 * (size_t)f;
#pragma unroll
      for (int k = 0; k < 6; ++k) xv[k] = xf[k];
    }
  }
  // frame data of wavefront 0: in flight while the W blocks arrive
  if (fin) {
    const double* fbp = fbuf + (size_t)f * MCBA_FB;
#pragma unroll
    for (int k = 0; k < 21; ++k) Lp[k] = fbp[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { gf[k] = fbp[27 + k]; D[k] = fbp[33 + k]; }
    frozen = (unsigned)fbp[39];   // coordinates a bound is active on (k_syrk's frame factor: mcba_set_frozen; 0 otherwise): their step is exactly 0
  }
  bool active;
  int sidx;
  double lambda;
  if (wait) {
    // ONE wavefront per workgroup polls (a thousand wavefronts hammering one memory channel delay the very store they wait
    // for), then fetches the camera step with cache-bypassing loads and hands it to the other wavefronts through LDS.  No cache
    // is invalidated: an agent-scope acquire in every wavefront costs an L2 invalidation each (measured: 6.2 us between "flag
    // seen" and "state read" with 118 of them queueing per XCD).
    double* mail = wait->mail;  // [0]: 1 camera step fetched, 0 no step this tick (or the solve never posted); [8 ..] camera step
    if (wave == 0) {
      const double base = 4.0 * wait->seq;
      // (base + 3: the solve ended without a step -- terminated, or nothing to solve; -1: the solve never posted -- leave the trial slot alone, and say so)
      const int rc = release_word_poll(wait->flag, base + 1.0, base + 2.0, base + 3.0, wait->max_polls);
      if (rc < 0 && lane == 0) backsub_stamp_timeout(wait);
      const int got = rc > 0 ? 1 : 0;
      release_word_acquired(wait->strict);
      if (got) {
        for (int i = lane; i < nc; i += 64) mail[8 + i] = load_coherent(wait->dc + i);
      }
      if (lane == 0) mail[0] = got ? 1.0 : 0.0;
    }
    __syncthreads();
    if (mail[0] == 0.0) return;
    FSTAMP(2);
    active = true;  // so far: the final word is awaited before anything is stored
    sidx = sidx_early;
    lambda = (wait->spec && wait->early[MCBA_LM_SKIP] == 0.0) ? lm_spec_lambda(wait->early[1], wait->lam_min, wait->dec_floor) : wait->early[1];
  } else {
    active = sel_active(sl, true);
    sidx = active ? sel_index(sl) : 0;  // current slot / linearisation; the trial goes to the other slot
    lambda = active ? sel_lambda(sl) : 0.0;
  }
  if (!active) return;
  FSTAMP(3);
  const double* __restrict__ rec = sidx ? rec1 : rec0;
  const double* xs = sidx ? x1 : x0;
  double* xd = sidx ? x0 : x1;
  const bool pre = wait != nullptr;  // the first camera's W block and the pose are already in registers
  auto dc = [&](int i) { return wait ? wait->mail[8 + i] : dcs.v[i]; };
  if (fin && !pre) {
    const double* xf = xs + n + 6 * (size_t)f;
#pragma unroll
    for (int k = 0; k < 6; ++k) xv[k] = xf[k];
  }
  double t[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int c = wave; c < C; c += nw) {
    // all 36 loads of the camera's W block in flight at once: a dependent round trip costs ~2 us, the data 0.1 us
    if (!(pre && c == wave)) {
      const double2* w2 = reinterpret_cast<const double2*>(rec + ((size_t)c * nfb + block) * (MCBA_REC * 64)) + lane;
#pragma unroll
      for (int k = 0; k < NR; ++k) v[k] = w2[(R0 + k) * 64];
    }
    double d[CW];
#pragma unroll
    for (int lr = 0; lr < CW; ++lr) d[lr] = dc(CW * c + lr);  // wave-uniform
#pragma unroll
    for (int lr = 0; lr < CW; ++lr) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { t[2 * k] = fma(v[3 * lr + k].x, d[lr], t[2 * k]); t[2 * k + 1] = fma(v[3 * lr + k].y, d[lr], t[2 * k + 1]); }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s_t[wave][k][lane] = t[k];
  }
  // (workgroup barrier among the wavefronts that got here: wavefronts past `nw` have ended, which the barrier accounts for)
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < nw; ++w) {  // fixed order: bit-reproducible
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] += s_t[w][k][lane];
  }
  double pred = 0.0, dn2 = 0.0, xn2 = 0.0;
  double xnew[6];
  if (fin) {
    double id[6], y[6], dl[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { id[k] = Lp[k * (k + 1) / 2 + k]; t[k] += gf[k]; }  // diagonal slots hold 1 / L_kk
    if (frozen) {
#pragma unroll
      for (int k = 0; k < 6; ++k) t[k] = ((frozen >> k) & 1u) ? 0.0 : t[k];
    }
    fwd6(Lp, id, t, y);
    bwd6(Lp, id, y, dl);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double d = -dl[k];
      xnew[k] = xv[k] + d;
      pred += d * (lambda * D[k] * d - gf[k]);
      dn2 += d * d;
      xn2 += xv[k] * xv[k];
    }
  }
  double a = wave_sum63(pred), b = wave_sum63(dn2), cc = wave_sum63(xn2);
  FSTAMP(4);
  if (wait) {  // the steps are ready; they count only if the solve's FINAL state still wants a trial step (a failed solve does not)
    const double fin_word = 4.0 * wait->seq + 2.0;
    const bool posted = release_word_poll(wait->flag, fin_word, fin_word, -1.0, wait->max_polls) > 0;
    release_word_acquired(wait->strict);
    if (!posted && lane == 0) backsub_stamp_timeout(wait);
    if (!posted || load_coherent(sl.lms + MCBA_LM_DONE) != 0.0 || load_coherent(sl.lms + MCBA_LM_SKIP) != 0.0) return;
  }
  if (block == 0) {
    if constexpr (CW == 12) {
      for (int i = lane; i < n; i += 64) xd[i] = xs[i] + dc(i);
    } else {
      for (int i = lane; i < n; i += 64) {
        const int cam = i / 12, l = i - 12 * cam;
        xd[i] = l >= 12 - CW ? xs[i] + dc(CW * cam + l - (12 - CW)) : xs[i];
      }
    }
  }
  if (fin) {
    double* xo = xd + n + 6 * (size_t)f;
#pragma unroll
    for (int k = 0; k < 6; ++k) xo[k] = xnew[k];
  }
  if (lane == 63) { bpart[3 * block] = a; bpart[3 * block + 1] = b; bpart[3 * block + 2] = cc; }
  FSTAMP(5);
}

}  // namespace mcba
