// mcba_solve.hip -- the reduced camera system solved ON THE GPU (BASELINE.json config 5: "on-GPU Schur solve").
//
// One workgroup factorises  S(lambda) = S0 + lambda diag(D_c)  (n = 12 C <= 480, FP64) and produces the camera step
// d_c = S^-1 rhs that k_backsub consumes straight from device memory, so that a Levenberg-Marquardt iteration needs
// no host round trip at all (solver.py `reduced_solver="device"`).  It replaces the host LAPACK call of
// solver._solve_spd, which itself stands where the reference calls LSMR (scipy/optimize/_lsq/trf.py:479-480).
//
// Algorithm: blocked Cholesky, block 16, of the matrix augmented with the right-hand side as ROW n
//            [ S   . ]        L y = rhs falls out of the factorisation as row n of the factor (no forward sweep)
//            [ rhs 1 ]
// Up to 9 cameras the factor lives in LDS (odd row stride) -- LEFT-LOOKING: the whole lower triangle is put in flight before
// anything else, because a dependent global round trip costs ~2 us here;
//   per block column k:   panel  = A[16k:, 16k:16k+16] - L[16k:, :16k] L[16k:16k+16, :16k]^T     v_mfma_f64_16x16x4, one
//                                                                                               wave per 16-row tile
//                         diagonal block AND the rows below it in ONE instruction stream: lanes 0..15 of every wavefront
//                         hold the 16 diagonal rows (redundantly), lanes 16..63 hold 48 further panel rows; finished
//                         entries of the diagonal rows are broadcast with v_readlane (left-looking inside the block), so
//                         there is no barrier between "factor" and "triangular solve"; pivots: v_rsq_f64 + one cubic step
//   backward sweep L^T d = y block by block: the identity's 16 rows ride through every diagonal block's triangular solve
//   in otherwise idle lanes, which leaves L_kk^-T -- the sweep's 16 sequential pivots per block become one 16-term dot
//   product per row (measured at 12 C = 72: 10.6 k -> 6.1 k cycles).
// Beyond that (12 C + 1 > 112 rows) the trailing matrix lives in an L2-resident scratch as 16 x 16 tiles in the MFMA
// accumulator layout -- RIGHT-LOOKING with a look-ahead panel: solve_right_looking below.
// k_solve_backsub (single-GPU ticks, factor in LDS) runs the back-substitution of the NEXT trial step in the same launch, in
// workgroups that wait for two words this workgroup releases (mcba_backsub.h).
// The same launch evaluates first-order optimality, applies the termination verdict (pending from k_sum_trial / k_decide,
// or taken here on the all-reduced trial scalars: one-collective ticks), handles a failed factorisation (more damping,
// next tick rebuild-only) and posts the LM state to a host-mapped ring slot.
#include <stdlib.h>
#include <algorithm>
#include "mcba_kernels.h"
#include "mcba_lm.h"
#include "mcba_math.h"
#include "mcba_backsub.h"

namespace mcba {

typedef double solve_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lane_bcast(double v, int lane) {  // `lane` wave-uniform
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
  b.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
  return b.d;
}

// 1/sqrt(a): hardware estimate + ONE third-order step  y (1 + e/2 + 3 e^2/8), e = 1 - a y^2  (estimate good to > 2^-20
// -> full FP64); shorter dependent chain than two Newton steps -- it sits on the critical path of every pivot
__device__ __forceinline__ double rsqrt_cubic(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double e = fma(-a * y, y, 1.0);
  double q = e * fma(0.375, e, 0.5);
  return fma(y, q, y);
}

// four values at once (sum or max), result to every thread: one pair of barriers instead of four
template <int NTHREADS>
__device__ __forceinline__ void block_reduce4(double (&v)[4], bool take_max, double* s_red) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double o = __shfl_xor(v[c], off, 64);
      v[c] = take_max ? fmax(v[c], o) : v[c] + o;
    }
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) s_red[c * 16 + wave] = v[c];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double r = s_red[c * 16];
    for (int w = 1; w < NTHREADS / 64; ++w) r = take_max ? fmax(r, s_red[c * 16 + w]) : r + s_red[c * 16 + w];
    v[c] = r;
  }
}

// LM state (LDS copy `st`) -> device state and the host-mapped ring slot; the sequence number goes last, after a
// system-scope fence, so that a host that sees it also sees the rest.  Called by every thread of the block.
__device__ __forceinline__ void post_state(const SolveArgs& a, const double* st, bool write_back, bool stepped = false) {
  const int tid = threadIdx.x;
  if (tid < MCBA_LMS - 1) {
    const double v = st[tid];
    if (write_back) a.lms[tid] = v;
    if (a.host_state) a.host_state[tid] = v;
  }
  if (!a.host_state && !a.flag) return;
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    if (a.host_state) *reinterpret_cast<volatile double*>(a.host_state + MCBA_LMS - 1) = a.seq;  // (the end of the kernel releases it)
    // k_solve_backsub: the back-substitution workgroups of the same launch are polling this word; the camera step (a.dc) and
    // the state were written by other threads of this workgroup before the fence and the barrier above
    // ONE word, three values per tick: 4 seq + 1 camera step in memory, 4 seq + 2 state final (after a step), 4 seq + 3 state
    // final and no step this tick -- a poll is one cache-bypassing load, not two
    if (a.flag) __hip_atomic_store(a.flag, 4.0 * a.seq + (stepped ? 2.0 : 3.0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#define STAMP(k) do { } while (0)

// look-ahead panel products (LDS-resident factor): one more npad x 17 buffer -- there is room for it up to 8 cameras next to
// k_solve_backsub's back-substitution scratch (160 KB per workgroup)
#ifndef MCBA_SOLVE_LOOKAHEAD
#define MCBA_SOLVE_LOOKAHEAD 1
#endif
__host__ __device__ inline bool solve_lookahead(int npad) { return MCBA_SOLVE_LOOKAHEAD && npad <= 96; }

// =====================================================================================================================
// RIGHT-LOOKING variant for systems that do not fit LDS (> 9 cameras; config 5: 12C = 288), 8 wavefronts.
//
// The left-looking form above reads, per block column, rows of the factor that grow with the column index from an L2-resident
// scratch: every block step pays several dependent global round trips (B rows -> LDS, A rows, the diagonal rows, the write
// back) -- 10.8 us per block step at 12C = 288 against 2.4 us with the factor in LDS.  Here every 16 x 16 tile of the trailing
// matrix is read and written ONCE per block step, in the MFMA accumulator layout (32 contiguous bytes per lane), the panel of
// the current step never leaves the workgroup (registers + LDS), and the next panel is produced first (look-ahead), so that the
// 16 sequential pivots of block k + 1 (wavefront 0) run while the other wavefronts finish the trailing update of block k.
//
// Tile layout: element (i, c) of a tile lives in lane i + 16 (c >> 2), register c & 3 -- a lane holds four CONTIGUOUS columns of
// its row, so a tile of the row-major reduce buffer is one aligned 32-byte load per lane.  The same registers are
//   * an MFMA B operand of the tile as it stands (lane (i, kk), K-step s  <->  column 4 kk + s: any K order serves as long as
//     both operands use it) and, read back from LDS with the lane's two 2-bit fields swapped (rl_pi), its A operand,
//   * the accumulator of the TRANSPOSED tile under that row permutation: D row 4 r + g of lane (i, g) is column 4 g + r,
// so  C^T -= L_J L_I^T  keeps the layout and a finished tile of the factor is an operand without any shuffle.  Per block step k
// (two barriers):
//   A  wavefront 0: the 16 pivots of the diagonal block, with the 16 identity rows riding in lanes 16..31 (-> L_kk^-T);
//      wavefronts 1-3, 5-7 (the SIMDs wavefront 0 is not on): trailing update with panel k - 1 of every tile right of column k,
//      a ring of tiles in flight, branch-free; at k = 0 they bring the tiles from the reduce buffer into the scratch instead;
//      wavefront 4 (same SIMD as the pivots, so no MFMAs): panel k - 1 from LDS to the tiles of the backward sweep
//   B  every wavefront, its tile rows I = wave, wave + 8, ...:  L_(I,k) = Z L_kk^-T  as four MFMAs on the registers the
//      look-ahead left (result to LDS: operands of A's updates), then at once the look-ahead -- its tiles of column k + 1
//      updated with panel k: the one operand it does not own, L_(k+1,k), every wavefront forms for itself; the diagonal tile
//      goes to LDS as rows for the pivots, the others stay in registers
// The backward sweep uses the inverse diagonal blocks; its rows of the factor are requested one step ahead.
// Measured (scripts/solve_time.py, one box): 12C = 120 / 192 / 288 / 480: 44 / 78 / 131 / 342 us against 117.6 (192) / 205 / 578 us
// left-looking; what is left at 288 (of ~300 k cycles; the first touch went to stager workgroups, rl_stager): B at ~4.5 k per step
// (a global round trip for the look-ahead tiles inside it: interleaving its MFMA chains changed nothing), the updates at 2 x their
// MFMA time (stamped per tile and wavefront: ~250 cycles of LDS operand reads, ~200 of MFMAs, ~160 of stores + the next request;
// with the MFMAs removed the interval is almost as long), the backward sweep 30 k.
#ifndef MCBA_RL_NB
#define MCBA_RL_NB 4
#endif
// a barrier that orders LDS traffic only: global loads issued before it stay in flight (__syncthreads waits for them)
__device__ __forceinline__ void rl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ size_t rl_tile(int I, int J) { return ((size_t)(I * (I + 1) / 2 + J)) << 8; }
// a tile in the scratch: registers (0, 1) of all lanes, then registers (2, 3) -- each 16-byte access of a wavefront is 1 KB of
// whole cache lines (with 32 contiguous bytes per lane every instruction touched half of 16 lines: first touch 45 k -> 38 k cycles)
typedef double solve_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ solve_d4 rl_tile_load(const double* t, int lane) {
  const solve_d2 lo = *reinterpret_cast<const solve_d2*>(t + 2 * lane), hi = *reinterpret_cast<const solve_d2*>(t + 128 + 2 * lane);
  return solve_d4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void rl_tile_store(double* t, int lane, const solve_d4 v) {
  *reinterpret_cast<solve_d2*>(t + 2 * lane) = solve_d2{v[0], v[1]};
  *reinterpret_cast<solve_d2*>(t + 128 + 2 * lane) = solve_d2{v[2], v[3]};
}
// lane a of an A operand carries row pi(a) of the tile (the 4 x 4 transpose of the lane's two 2-bit fields), see above
__device__ __forceinline__ int rl_pi(int lane) { return ((lane & 3) << 2) | ((lane >> 2) & 3) | (lane & 48); }

// tile (I, J), I >= J, of the augmented damped matrix straight from the reduce buffer (both triangles are there: k_reduce_system
// mirrors on write): lane (i, g) = row 16 I + i, columns 16 J + 4 g .. + 3 -- one aligned 32-byte load (n = 12 C: a group of four
// columns is either inside the matrix or past it)
__device__ __forceinline__ solve_d4 rl_system_load(const double* __restrict__ S0, int n, int I, int J, int lane) {
  const int row = 16 * I + (lane & 15), c0 = 16 * J + 4 * (lane >> 4);
  const bool in = row <= n && c0 < n;  // (row n of the stride-n array is the right-hand side)
  return *reinterpret_cast<const solve_d4*>(S0 + (in ? (size_t)row * n + c0 : 0));
}
// ... and what is not in the buffer: the damping on the diagonal, 1 at (n, n), the identity padding, parameters held fixed.
// (Kept apart from the load so that a caller can have many tiles in flight: the rare-path branch below ends the compiler's
//  static wait counts.)
__device__ __forceinline__ solve_d4 rl_system_fixup(solve_d4 v, const double* damp, const unsigned char* fixed, int n, int I, int J, int lane) {
  const int row = 16 * I + (lane & 15), c0 = 16 * J + 4 * (lane >> 4);
  const bool in = row <= n && c0 < n;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + r;
    double x = in ? v[r] : ((row == c) ? 1.0 : 0.0);
    if (in && row == c) x += damp[row];
    v[r] = x;
  }
  if (fixed) {  // identity row and column, zero right-hand side
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + r;
      const bool fr = row < n && fixed[min(row, n - 1)] != 0, fc = c < n && fixed[min(c, n - 1)] != 0;
      if (fr || fc) v[r] = (row == c) ? 1.0 : 0.0;
    }
  }
  return v;
}
__device__ __forceinline__ solve_d4 rl_tile_from_system(const double* __restrict__ S0, const double* damp, const unsigned char* fixed, int n, int I, int J, int lane) {
  return rl_system_fixup(rl_system_load(S0, n, I, J, lane), damp, fixed, n, I, J, lane);
}

// C^T -= L_J L_I^T on one tile: `pj` is the panel's tile of rows J read as an A operand (rl_pi), `pi` its tile of rows I as it stands
__device__ __forceinline__ solve_d4 rl_update(solve_d4 c, const solve_d4 pj, const solve_d4 pi) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pj[s], pi[s], c, 0, 0, 0);
  return c;
}

// flat index t of a lower-triangular enumeration (row-major) -> (row, column)
__device__ __forceinline__ void rl_unflatten(int t, int& Ir, int& Jr) {
  Ir = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((Ir + 1) * (Ir + 2) / 2 <= t) ++Ir;
  while (Ir * (Ir + 1) / 2 > t) --Ir;
  Jr = t - Ir * (Ir + 1) / 2;
}

// ---- stager workgroups (blockIdx.x >= 1 of the right-looking variant's launch).  The reduce buffer was written by other XCDs and
// ONE compute unit pulls it at ~45 GB/s (0.66 MB at 24 cameras: 38 k cycles, 2 MB at 40 cameras: 99 k -- more tiles in flight per
// wavefront change nothing), so the OFF-DIAGONAL tiles right of column 0 -- no damping in them, nothing that depends on the LM
// state -- are brought into the tile scratch by NS other workgroups, a tile or two per wavefront, while workgroup 0 reads its
// state, takes its decision and does the first 16 pivots.  Protocol (T = the launch's number in the handle's life, SolveArgs.stage_tag;
// the scratch is zero at allocation, T only grows):
//   stager b:     CLAIMS its share first -- atomic max of its claim word with 2 T: an old value below 2 T means "mine", 2 T + 1 means
//                 workgroup 0 has given up on it: it exits WITHOUT a store --, stages its tiles, releases its done word = T.
//   workgroup 0:  waits (bounded) for the done words; for a stager whose word has not come it raises the claim word to 2 T + 1: an
//                 old value of 2 T means the stager is running -- its word is then awaited without the bound --, anything lower means
//                 it never started and now never will: workgroup 0 stages that share itself.
// (Round 3 let workgroup 0 re-stage everything after the bounded wait with no claim: a stager that was only LATE -- an oversubscribed
//  GPU -- would then have written the original tiles over ones the factorisation had already updated.  ADVICE r3.)
constexpr int kRlMaxStagers = 16;  // (8, 16 and 32 measured the same to 1 %: 12C = 288: 129.2 / 129.6 / 131.4 us)
__host__ __device__ inline int rl_stagers(int npad) { const int nblk = npad >> 4, moff = (nblk - 1) * (nblk - 2) / 2; return min(kRlMaxStagers, max(1, moff / 8)); }
__device__ __forceinline__ double* rl_stage_flags(const SolveArgs& a, int nblk) { return a.work + 2 * rl_tile(nblk, 0) + 256 * 8; }
__device__ __forceinline__ unsigned long long* rl_stage_claims(const SolveArgs& a, int nblk) { return reinterpret_cast<unsigned long long*>(rl_stage_flags(a, nblk) + kRlMaxStagers); }
__device__ __forceinline__ unsigned long long rl_stage_tag(const SolveArgs& a) { return (unsigned long long)fabs(a.stage_tag); }  // (test hooks: a negative tag, a tag + 0.5)
struct RlNoGate { __device__ __forceinline__ bool operator()() const { return true; } };
// `gate` is called once, after the first batch of loads has been issued and before the first store: false = leave without storing
template <class Gate = RlNoGate>
__device__ __forceinline__ void rl_stage_offdiag(const SolveArgs& a, int part, int nparts, int lane, int wave, Gate gate = Gate()) {  // tiles (I, J), 1 <= J < I, share `part` of `nparts`
  const int n = a.n, nblk = a.npad >> 4, m = nblk - 2;
  if (m <= 0) { (void)gate(); return; }
  const int M = m * (m + 1) / 2;
  const int t_begin = (int)(((long long)M * part) / nparts), ntile = (int)(((long long)M * (part + 1)) / nparts) - t_begin;
  if (ntile <= 0) (void)gate();  // (every wavefront of a stager passes the gate's barrier exactly once)
  const double* __restrict__ S0 = a.red;
  double* T1 = a.work;
  int Ir, Jr;
  rl_unflatten(t_begin, Ir, Jr);
  for (int done = 0; done < ntile; done += 8) {
    solve_d4 v[8];
    double* dstp[8];
    int Is[8], Js[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = done + j < ntile;
      Is[j] = ok ? Ir + 2 : nblk - 1; Js[j] = ok ? Jr + 1 : nblk - 2;
      dstp[j] = ok ? T1 + rl_tile(Is[j], Js[j]) : a.work + 2 * rl_tile(nblk, 0) + 256 * wave;
      v[j] = rl_system_load(S0, n, Is[j], Js[j], lane);
      if (++Jr > Ir) { ++Ir; Jr = 0; }
    }
    if (done == 0 && !gate()) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) rl_tile_store(dstp[j], lane, rl_system_fixup(v[j], nullptr, a.fixed, n, Is[j], Js[j], lane));  // (no diagonal entry in these tiles: no damping)
  }
}
__device__ __forceinline__ void rl_stager(const SolveArgs& a, int b, int ns) {
  if (a.stage_tag < 0.0) return;  // (MCBA_SOLVE_STAGERS=-1, tests: stagers that never show up -- workgroup 0 must time out and do the work itself)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  // (the claim's outcome travels through the first word of the DYNAMIC LDS, which a stager does not use otherwise: a static __shared__
  //  variable in this kernel moves the dynamic region off its 16-byte alignment and every 32-byte LDS access of workgroup 0's block
  //  steps with it -- 130 -> 222 us at 12C = 288 when tried)
  extern __shared__ double smem[];
  volatile int* s_mine_p = reinterpret_cast<volatile int*>(smem);
#define s_mine (*s_mine_p)
  // the claim (an atomic round trip, ~2 us) is taken while the first tiles are already on their way: the gate sits between the loads and the
  // first store
  auto gate = [&]() -> bool {
    if (threadIdx.x == 0) {
      if (a.stage_tag != floor(a.stage_tag)) {  // (MCBA_SOLVE_STAGERS=-2, tests: a LATE stager -- it shows up ~0.2 s after workgroup 0 has given up on it)
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 20000000LL) __builtin_amdgcn_s_sleep(64);
      }
      const unsigned long long tag2 = 2 * rl_stage_tag(a);
      const unsigned long long old = __hip_atomic_fetch_max(rl_stage_claims(a, a.npad >> 4) + b, tag2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_mine = old < tag2 ? 1 : 0;
    }
    __syncthreads();
    return s_mine != 0;  // false: workgroup 0 gave up on this stager and does (or did) its share -- not a single store from here
  };
  // (the dummy tile of a wavefront past its run: the stagers share workgroup 0's eight -- harmless, nobody reads them)
  rl_stage_offdiag(a, b * nw + wave, ns * nw, lane, wave & 7, gate);
  if (!s_mine) return;
#undef s_mine
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(rl_stage_flags(a, a.npad >> 4) + b, a.stage_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NTHREADS>
__device__ __forceinline__ void solve_right_looking(const SolveArgs& a, const double* __restrict__ S0, const unsigned char* fixed, const double* damp, double* yv, double* dv, double* invd,
                                                    double* linv /* npad x 17 */, double* Pop /* npad x 16 */, double* dtile /* 16 x 17 */, double* znext /* 2 x 256 */, int* stage_miss /* LDS word */, double* tstamp) {
  constexpr int NW = NTHREADS / 64, SL = 32 / NW;  // tile row I is owned by wavefront I mod NW (<= 32 tile rows: 40 cameras)
  static_assert(NW == 8, "8 wavefronts: 256 VGPRs each");
  constexpr int NWK = NW - NW / 4;                // trailing-update workers: the wavefronts that do not share a SIMD with wavefront 0 (the pivots)
  const int n = a.n, npad = a.npad, nblk = npad >> 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform, and the compiler knows it: tile indices and addresses stay in SGPRs)
  const bool worker = (wave & 3) != 0;
  const int wk = wave - 1 - (wave >> 2);                      // worker index 0 .. NWK - 1
  const int li = lane & 15, lg = lane >> 4, lpi = rl_pi(lane);
  double* T1 = a.work;                                        // trailing tiles
  double* T2 = a.work + rl_tile(nblk, 0);                     // the factor, column-major tiles (backward sweep)
  for (int j = tid; j < npad; j += NTHREADS) yv[j] = 0.0;
  solve_d4 Zp[SL];                                            // my tiles of the current panel (rows wave, wave + NW, ...)
#pragma unroll
  for (int sl = 0; sl < SL; ++sl) {
    const int I = wave + NW * sl;
    Zp[sl] = solve_d4{0.0, 0.0, 0.0, 0.0};
    if (I >= 1 && I < nblk) Zp[sl] = rl_tile_from_system(S0, damp, fixed, n, I, 0, lane);
    if (I == 1 && I < nblk) *reinterpret_cast<solve_d4*>(znext + 4 * lane) = Zp[sl];  // tile (k + 1, k) of the coming step, for everybody
  }
  if (wave == 0) {
    const solve_d4 v = rl_tile_from_system(S0, damp, fixed, n, 0, 0, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) dtile[li * 17 + 4 * lg + r] = v[r];
  }
  __syncthreads();
  for (int k = 0; k < nblk; ++k) {
    const int r0 = 16 * k;
    // ---- A: pivots of block k (wavefront 0)  ||  the others: trailing update with panel k - 1 of the columns >= k + 1
    if (wave == 0) {
      double r[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) r[c] = lane < 16 ? dtile[li * 17 + c] : ((lane < 32 && c == li) ? 1.0 : 0.0);
      double myinv = 1.0;
      // (round 6) the block that holds the right-hand-side row is the last one: its rows from row n on are the right-hand side (not a pivot)
      // and identity padding -- columns whose elimination step multiplies by 1 and subtracts 0.  The steps j >= n - r0 are skipped: exact
      // (bit-identical), and they are the longest ones of the block (step j is j broadcasts + FMAs deep)
      const int jmax = r0 + 16 <= n ? 16 : n - r0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < jmax) {   // (wave-uniform; a `break` here would keep the loop from unrolling and r[] from living in registers)
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int qq = 0; qq < j; ++qq) {
            const double ljq = lane_bcast(r[qq], j);
            if (qq & 1) s1 = fma(r[qq], ljq, s1); else s0 = fma(r[qq], ljq, s0);
          }
          r[j] -= s0 + s1;
          const double pj = lane_bcast(r[j], j);
          const double inv = rsqrt_cubic(pj);
          r[j] *= inv;
          myinv = lane == j ? inv : myinv;
        }
      }
      if (lane < 16) {
        invd[r0 + lane] = myinv;
        if (r0 + lane == n) {  // the right-hand-side row ends in this block: y for the block's columns left of it
#pragma unroll
          for (int c = 0; c < 16; ++c) if (c < lane) yv[r0 + c] = r[c];
        }
      } else if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 16; ++c) linv[(r0 + li) * 17 + c] = r[c];
      }
    } else {
    // wavefront 4 (shares its SIMD with the pivots): first the finished panel k - 1, still in Pop, to the tiles of the backward
    // sweep -- four 8-byte LDS reads and one 32-byte store per tile and lane (lane (c, g): rows 4 g .. + 3) --, then, while the
    // updates are the longer side of the interval anyway, a seventh share of them (its MFMAs slow the pivots down, which then
    // does not matter); on the short steps it leaves the SIMD to the pivots
    const int mleft = nblk - k - 1;
    const int nwk = k == 0 || mleft * (mleft + 1) / 2 >= 48 ? NWK + 1 : NWK;
    const int wq = worker ? wk : NWK;  // this wavefront's share
    if (!worker && k > 0) {
      for (int I = k; I < nblk; ++I) {
        solve_d4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = Pop[256 * I + 4 * ((4 * lg + r) + 16 * (li >> 2)) + (li & 3)];
        *reinterpret_cast<solve_d4*>(T2 + rl_tile(I, k - 1) + 64 * lg + 4 * li) = x;  // [row quad][column][row in quad]: threads of one column group read 32-byte neighbours
      }
    }
    if (wq >= nwk) {
      // (wavefront 4 on a short step)
    } else if (k == 0) {
      // first touch: the DIAGONAL tiles right of column 0 (they carry the damping) from the reduce buffer into the tile scratch, while
      // wavefront 0 does the first 16 pivots; the off-diagonal ones come from the stager workgroups (rl_stager)
      {
        solve_d4 v[4];
        int Id[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Id[j] = 1 + wq + nwk * j;
          v[j] = rl_system_load(S0, n, min(Id[j], nblk - 1), min(Id[j], nblk - 1), lane);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (Id[j] < nblk) rl_tile_store(T1 + rl_tile(Id[j], Id[j]), lane, rl_system_fixup(v[j], damp, fixed, n, Id[j], Id[j], lane));
        }
        for (int Ix = 1 + wq + nwk * 4; Ix < nblk; Ix += nwk)  // (more than 29 tile rows: the rest one at a time)
          rl_tile_store(T1 + rl_tile(Ix, Ix), lane, rl_tile_from_system(S0, damp, fixed, n, Ix, Ix, lane));
      }
      if (wave == 1) {  // ... and one wavefront waits for the stagers' words (lane b: stager b), bounded
        const int ns = (int)gridDim.x - 1;
        const double* fl = rl_stage_flags(a, nblk);
        const int lb = min(lane, kRlMaxStagers - 1);
        bool ok = lane >= ns;
        for (int polls = 0; polls < 40000 && !__all(ok); ++polls) {
          if (!ok) ok = __hip_atomic_load(fl + lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fabs(a.stage_tag);
          if (!__all(ok)) __builtin_amdgcn_s_sleep(8);
        }
        bool abandoned = false;
        if (!ok) {  // this stager's word has not come: take its share away from it, unless it is already at work
          const unsigned long long tag2 = 2 * rl_stage_tag(a);
          const unsigned long long old = __hip_atomic_fetch_max(rl_stage_claims(a, nblk) + lb, tag2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == tag2) {  // it has started, so it will finish: its word is awaited without the bound
            while (__hip_atomic_load(fl + lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fabs(a.stage_tag)) __builtin_amdgcn_s_sleep(8);
          } else {
            abandoned = true;
          }
        }
        const unsigned long long gone = __ballot(abandoned);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // what the stagers wrote is visible to this CU from here on
        if (lane == 0) *stage_miss = (int)(gone & 0xFFFFu);   // the stagers whose share this workgroup stages itself
      }
    } else if (k + 1 < nblk) {
      const int J0 = k + 1, m = nblk - J0;        // panel k - 1 is in Pop; tiles (I, J), J0 <= J <= I < nblk: m (m + 1) / 2 of them
      const int M = m * (m + 1) / 2;
      const int t_begin = (int)(((long long)M * wq) / nwk), ntile = (int)(((long long)M * (wq + 1)) / nwk) - t_begin;
      if (ntile > 0) {
        // A ring of R tiles, branch-free: slot j multiplies and stores its tile, then requests the tile R places further on into the
        // same registers -- R - 1 tile updates (~300 cycles each) cover the round trip.  Past the end of this wavefront's run the
        // slots work on a dummy tile of its own (harmless operands, a scratch destination), so that every wait count is static.
        constexpr int R = MCBA_RL_NB;
        double* const dummy = a.work + 2 * rl_tile(nblk, 0) + 256 * wave;
        int Ir, Jr, issued = 0;
        rl_unflatten(t_begin, Ir, Jr);
        solve_d4 cin[R];
        int It[R], Jt[R];
        double* dst[R];
        auto request = [&](int j) {
          const bool ok = issued < ntile;
          It[j] = ok ? Ir + J0 : nblk - 1; Jt[j] = ok ? Jr + J0 : nblk - 1;
          double* tp = T1 + rl_tile(It[j], Jt[j]);
          dst[j] = ok ? tp : dummy;
          cin[j] = rl_tile_load(tp, lane);
          ++issued;
          if (++Jr > Ir) { ++Ir; Jr = 0; }
        };
#pragma unroll
        for (int j = 0; j < R; ++j) request(j);
        // (the operands of the NEXT slot are read from LDS before this slot's MFMAs: stamped per tile, the four ds_read_b128 +
        //  their wait were 250 cycles against 200 for the MFMAs and 160 for the stores and the next request)
        solve_d4 pi_n = *reinterpret_cast<const solve_d4*>(Pop + 256 * It[0] + 4 * lane);
        solve_d4 pj_n = *reinterpret_cast<const solve_d4*>(Pop + 256 * Jt[0] + 4 * lpi);
        for (int done = 0; done < ntile; done += R) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const solve_d4 pi = pi_n, pj = pj_n;
            const int jn = (j + 1) % R;   // (slot 0 of the next trip has been requested already)
            pi_n = *reinterpret_cast<const solve_d4*>(Pop + 256 * It[jn] + 4 * lane);
            pj_n = *reinterpret_cast<const solve_d4*>(Pop + 256 * Jt[jn] + 4 * lpi);
            const solve_d4 out = rl_update(cin[j], pj, pi);
            rl_tile_store(dst[j], lane, out);
            request(j);
          }
        }
      }
    }
    }
    __syncthreads();
    if (k == 0 && ((int)gridDim.x == 1 || *stage_miss)) {
      // no stagers in this launch: the same tiles, by this workgroup; or stagers that never started (their claim words now say so: they
      // will not write): their shares
      const int ns = (int)gridDim.x - 1, gone = ns == 0 ? 1 : *stage_miss, nsh = ns == 0 ? 1 : ns;
      for (int b = 0; b < nsh; ++b)
        if ((gone >> b) & 1) rl_stage_offdiag(a, b * NW + wave, nsh * NW, lane, wave);
      __syncthreads();
    }
    // ---- B: my tiles of panel k:  L_(I,k) = Z L_kk^-T,  then the look-ahead at once: my tiles of column k + 1 updated with panel k.
    // The one operand that is not mine -- L_(k+1,k) -- every wavefront forms for itself (four MFMAs on the tile its owner left in
    // LDS a step ago, read with the A operand's lane order): no barrier between the two.
    {
      solve_d4 cpre[SL];
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int I = wave + NW * sl;
        cpre[sl] = solve_d4{0.0, 0.0, 0.0, 0.0};
        if (I >= k + 1 && I < nblk) cpre[sl] = rl_tile_load(T1 + rl_tile(I, k + 1), lane);
      }
      solve_d4 lv;  // A operand of  X^T = L_kk^-1 Z^T : lane (a, kk), register s:  L_kk^-1[pi(a)][4 kk + s] = L_kk^-T[4 kk + s][pi(a)]
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) lv[s2] = linv[(r0 + 4 * lg + s2) * 17 + (lpi & 15)];
      solve_d4 pja = {0.0, 0.0, 0.0, 0.0};  // L_(k+1,k) as an A operand
      if (k + 1 < nblk) {
        const solve_d4 zn = *reinterpret_cast<const solve_d4*>(znext + 256 * (k & 1) + 4 * lpi);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) pja = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[s2], zn[s2], pja, 0, 0, 0);
      }
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int I = wave + NW * sl;
        if (I > k && I < nblk) {
          solve_d4 xo = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) xo = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[s2], Zp[sl][s2], xo, 0, 0, 0);
          *reinterpret_cast<solve_d4*>(Pop + 256 * I + 4 * lane) = xo;
          if (16 * I + li == n) {  // the right-hand-side row: y for this block's columns
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r0 + 4 * lg + r] = xo[r];
          }
          const solve_d4 c = rl_update(cpre[sl], pja, xo);   // tile (I, k + 1)
          if (I == k + 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dtile[li * 17 + 4 * lg + r] = c[r];
          } else {
            Zp[sl] = c;
            if (I == k + 2) *reinterpret_cast<solve_d4*>(znext + 256 * ((k + 1) & 1) + 4 * lane) = c;
          }
        }
      }
    }
    __syncthreads();
  }
  // (the last panel has no tiles below its diagonal block: nothing is left in Pop to bring to T2)
  // ---- backward sweep  L^T d = y  with the inverse diagonal blocks; thread j owns column j.  The rows of the factor a step needs
  // (block row k, column j) are requested one step ahead: a round trip to the scratch costs more than the step's arithmetic.
  {
    solve_d4 wc[4], wn[4];  // wc[q][r] = L[16 k + 4 q + r][tid]
    auto fetch = [&](int kk, solve_d4 (&w)[4]) {
      const bool ok = kk >= 1 && tid < 16 * kk;
      const double* src = T2 + rl_tile(ok ? kk : 1, ok ? (tid >> 4) : 0) + 4 * (tid & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const solve_d4*>(src + 64 * q);
    };
    auto step = [&](int k, solve_d4 (&w)[4], solve_d4 (&wnext)[4]) {  // block step k with its rows in `w`; requests step k - 1 into `wnext`
      const int r0 = 16 * k;
      fetch(k - 1, wnext);
      if (tid < 16) {
        const double* lr = linv + (r0 + tid) * 17;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 16; ++c) s4[c & 3] = fma(lr[c], yv[r0 + c], s4[c & 3]);
        dv[r0 + tid] = (r0 + tid < n) ? (s4[0] + s4[1]) + (s4[2] + s4[3]) : 0.0;
      }
      rl_lds_barrier();
      if (tid < r0) {
        double s0 = yv[tid], s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s0 = fma(-w[q][0], dv[r0 + 4 * q], s0);
          s1 = fma(-w[q][1], dv[r0 + 4 * q + 1], s1);
          s0 = fma(-w[q][2], dv[r0 + 4 * q + 2], s0);
          s1 = fma(-w[q][3], dv[r0 + 4 * q + 3], s1);
        }
        yv[tid] = s0 + s1;
      }
      rl_lds_barrier();
    };
    fetch(nblk - 1, wc);
    for (int k = nblk - 1; k >= 0; k -= 2) {  // two steps per trip: the two register sets swap roles without a copy
      step(k, wc, wn);
      if (k >= 1) step(k - 1, wn, wc);
    }
  }
}

// Staging of the reduce buffer: thread (rr, cc) = (tid / 16, tid % 16) owns the elements (rr + RS a, cc + 16 b) --
// all index arithmetic is incremental, every load is unconditional (clamped address) so none of them waits for another.
// KS x KS elements per thread cover npad <= 16 KS with 256 threads (KS = 5: up to 6 cameras, KS = 7: up to 9).
constexpr int kStageMax = 7;

template <int NTHREADS, bool LDSW, int KS>
__device__ __forceinline__ void solve_cam_body(const SolveArgs& a) {
  constexpr int kStage = KS;
  extern __shared__ double smem[];
  const int n = a.n, npad = a.npad, nblk = npad >> 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NW = NTHREADS / 64;
  constexpr int RS = NTHREADS / 16;                    // rows per staging pass
  double* panel = smem;                                // npad x 17 : sum_p L[i][p] L[r0+c][p] of the current panel
  double* yv = panel + (size_t)npad * 17;              // npad : y, then scratch of the backward sweep
  double* dv = yv + npad;                              // npad : d
  double* invd = dv + npad;                            // npad : 1 / L_ii
  double* damp = invd + npad;                          // npad : lambda * D_c
  double* s_red = damp + npad;                         // 4 x 16 + 8
  int* s_flag = reinterpret_cast<int*>(s_red + 64);    // [0] mode
  double* lst = s_red + 72;                            // MCBA_LMS : the LM state, worked on in LDS
  double* W = lst + MCBA_LMS;                          // LDSW: npad rows, row-major: the factor L (lower part)
  const int ldw = npad + 1;                            // odd row stride in LDS: rows land in different banks
  double* linv = W + (size_t)npad * ldw;               // LDSW: nblk x 16 x 17, the inverse transposes of the diagonal blocks of L
  double* pnext = linv + (size_t)nblk * 16 * 17;       // LDSW, look-ahead: npad x 17, the next panel's products with the columns that are already final
  constexpr int kLookWave = 2;                         // the wavefront that is idle in the diagonal phase (NW = 4, <= 7 tiles)
  const bool lookahead = LDSW && NW == 4 && solve_lookahead(npad);
  double* Bs = lst + MCBA_LMS;                         // !LDSW: the right-looking variant's LDS (16 (npad + 2) doubles: the panel as MFMA operands; + 16 x 17 + 2 x 256 behind)
  const int bst = npad + 2;

  const double* __restrict__ S0 = a.red;               // rhs follows S0: it is "row n" of the same stride-n array
  const double* __restrict__ diagU = a.red + (size_t)n * n + n;
  const double* __restrict__ gc = diagU + n;
  const double* __restrict__ scal = gc + n;
  const unsigned char* fixed = a.fixed;
  const int rr = tid >> 4, cc = tid & 15;

  // ---- a dependent global round trip costs ~2 us here (the reduce buffer was written by other XCDs): with the factor
  // in LDS the whole lower triangle goes in flight before anything else
  double stage[LDSW ? kStage * kStage : 1];
  if (LDSW) {
#pragma unroll
    for (int ia = 0; ia < kStage; ++ia) {
      const int i = rr + RS * ia;
#pragma unroll
      for (int b = 0; b < kStage; ++b) {
        const int j = cc + 16 * b;
        const bool need = i <= n && j < n && j <= i;
        stage[ia * kStage + b] = S0[need ? i * n + j : 0];
      }
    }
  }

  // ---- LM state -> LDS; first-order optimality of the CURRENT point (the reduced system was built there)
  if (tid < MCBA_LMS) lst[tid] = a.lms_in[tid];
  double gm = 0.0;
  for (int i = tid; i < npad; i += NTHREADS) {
    const bool in = i < n, fx = in && fixed && fixed[i];
    const double dsc = (in && a.dscale) ? a.dscale[a.cw == 12 ? i : 12 * (i / 6) + 6 + i % 6] : 0.0;  // numeric x_scale: the caller's fixed D = 1 / x_scale^2 (in the layout of x); 0 = Marquardt's diag(U) for this parameter
    const double du = in ? (dsc > 0.0 ? dsc : diagU[i]) : 1.0;
    damp[i] = du > 0.0 ? du : 1.0;  // scaled by lambda below (the state is not in LDS yet)
    if (in && !fx) gm = fmax(gm, fabs(gc[i]));
  }
  if (tid < 12) gm = fmax(gm, scal[4 + tid]);
  const double frame_fail = scal[2];
  double red4[4] = {gm, 0.0, 0.0, 0.0};
  block_reduce4<NTHREADS>(red4, true, s_red);
  const double g_inf = red4[0];
  double* lms = lst;
  if (lms[MCBA_LM_DONE] != 0.0) {  // already terminated: only acknowledge the tick
    post_state(a, lst, false);
    return;
  }
  if (tid == 0) {
    int mode = 0;  // 0 solve, 1 terminated, 2 no solve possible (a frame block failed to factorise), 3 mispredicted
    lms[MCBA_LM_TICK] += 1.0;
    if (a.decide) {  // one-collective ticks: the decision is taken here, on the all-reduced trial scalars
      if (lms[MCBA_LM_SKIP] != 0.0) lm_mark_rebuild(lms);  // this tick only rebuilt the system (no trial)
      else if (a.red[(size_t)n * n + 3 * n + 16 + 5] != 0.0) {
        // the previous tick's fused back-substitution gave up waiting ON SOME SHARD (trial scalar 5: k_reduce_system's flag, summed by
        // the collective -- a shard that found out from its own word alone would part from the others here): the trial point (and
        // the speculative reduction built on it) is stale -- nothing is decided, the next tick rebuilds the system of the current point
        lm_mark_rebuild(lms);
        lms[MCBA_LM_SKIP] = 1.0;
        lms[MCBA_LM_SOLVE_INFO] = 4.0;
        mode = 3;
      } else {
        const double lam_spec = lm_spec_lambda(lms[1], a.lam_min, a.dec_floor);  // what the Schur reduction before us assumed
        lm_decide(a.red + (size_t)n * n + 3 * n + 16, DecideArgs{2, 0.0, 0.0, 0.0, a.lam_min, a.lam_max, lms, a.ftol, a.xtol, a.dec_floor});
        if (!(lms[4] != 0.0 && lms[1] == lam_spec)) {  // rejected, or accepted with another damping: the system in the
          lms[MCBA_LM_SKIP] = 1.0;                     // buffer is not the one to solve -> the next tick rebuilds it
          lms[MCBA_LM_SOLVE_INFO] = 3.0;
          mode = 3;
        }
      }
    }
    if (mode == 0) {
      lms[MCBA_LM_GINF] = g_inf;
      const double pending = lms[MCBA_LM_PENDING];
      if (pending != 0.0) { lms[MCBA_LM_DONE] = pending; mode = 1; }
      else if (g_inf < a.gtol) { lms[MCBA_LM_DONE] = 1.0; mode = 1; }
      else if (frame_fail != 0.0) mode = 2;
    }
    s_flag[0] = mode;
  }
  __syncthreads();
  const int mode = s_flag[0];
  if (mode == 1 || mode == 3) {
    post_state(a, lst, true);
    return;
  }
  STAMP(0);
  const double lambda = lms[1];
  const int sel = static_cast<int>(lms[3]) & 1;
  const double* xc = sel ? a.x1 : a.x0;

  if (mode == 0) {
    // ---- augmented, damped matrix -> W (lower triangle; the strictly upper part is never read as data):
    //   rows < n: S0 + lambda D_c on the diagonal,  row n: the right-hand side, 1 on the diagonal,  rows > n: identity
    for (int i = tid; i < npad; i += NTHREADS) damp[i] *= lambda;
    __syncthreads();
    if constexpr (!LDSW) {
      static_assert(NTHREADS == 512, "the right-looking variant is written for 8 wavefronts");
      STAMP(1);
      solve_right_looking<NTHREADS>(a, S0, fixed, damp, yv, dv, invd, panel, Bs, Bs + (size_t)16 * bst, Bs + (size_t)16 * bst + 16 * 17, s_flag + 2, lst);
    } else {
    {
#pragma unroll
      for (int ia = 0; ia < kStage; ++ia) {
        const int i = rr + RS * ia;
#pragma unroll
        for (int b = 0; b < kStage; ++b) {
          const int j = cc + 16 * b;
          if (i <= n && j < n && j <= i) W[i * ldw + j] = stage[ia * kStage + b] + ((i == j) ? damp[i] : 0.0);
        }
      }
    }
    for (int e = n * npad + n + tid; e < npad * npad; e += NTHREADS) {  // W[n][n] = 1 and the identity rows of the padding
      const int i = e / npad, j = e - i * npad;
      if (j <= i && (i > n || j == n)) W[(size_t)i * ldw + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (fixed) {  // parameters held fixed: identity row and column, zero right-hand side (uniform branch, rare path)
      for (int e = tid; e < (n + 1) * npad; e += NTHREADS) {
        const int i = e / npad, j = e - i * npad;
        if (j < n && j <= i && ((i < n && fixed[i]) || fixed[j])) W[(size_t)i * ldw + j] = (i == j) ? 1.0 : 0.0;
      }
      __syncthreads();
    }
    // Row lanes: lanes 0..15 of EVERY wavefront hold the 16 rows of the diagonal block (redundantly -- their finished
    // entries are what v_readlane broadcasts), lanes 16..63 hold 48 further rows of the panel each.
    const int q = lane < 16 ? lane : 16 + 48 * wave + (lane - 16);  // row inside the panel
    STAMP(1);
#define LAP(i) do { } while (0)
    // ---- factorisation
    for (int k = 0; k < nblk; ++k) {
      const int r0 = 16 * k, ntile = nblk - k;
      if (k > 0) {
        {
          // panel update, one 16x16 tile per wavefront pass.  With look-ahead (below) only the K-chunk of the block column
          // factorised LAST remains to be multiplied here: the products with all earlier columns were formed during that block's
          // diagonal phase and wait in `pnext`.
          const int p_lo = lookahead ? r0 - 16 : 0;
          for (int t = wave; t < ntile; t += NW) {
            const int R = r0 + 16 * t;
            solve_d4 acc = {0.0, 0.0, 0.0, 0.0};
            if (lookahead && k > 1) {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) acc[reg] = pnext[(16 * t + 4 * reg + (lane >> 4)) * 17 + (lane & 15)];
            }
            const double* pa = W + (size_t)(R + (lane & 15)) * ldw + (lane >> 4);
            const double* pb = W + (size_t)(r0 + (lane & 15)) * ldw + (lane >> 4);
            for (int p = p_lo; p < r0; p += 16) {  // r0 is a multiple of 16: four operand pairs in flight
              double av[4], bv[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) { av[u] = pa[p + 4 * u]; bv[u] = pb[p + 4 * u]; }
#pragma unroll
              for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) panel[(16 * t + 4 * reg + (lane >> 4)) * 17 + (lane & 15)] = acc[reg];
          }
        }
        __syncthreads();
      }
      LAP(0);
      // With the factor in LDS the last 16 lanes of the last wavefront (row slots no matrix of this variant reaches) carry
      // the rows of the 16 x 16 IDENTITY through the same triangular solve: what comes out is L_kk^-T, which turns the
      // 16 sequential pivots of this block in the backward sweep into one 16-term dot product per row.
      const bool idrow = LDSW && wave == NW - 1 && lane >= 48;
      if (lookahead && wave == kLookWave && k >= 1 && k + 1 < nblk) {
        // LOOK-AHEAD: this wavefront has no rows in the diagonal phase (<= 7 tiles: wavefronts 0, 1 and the last one do) -- while
        // wavefront 0 walks the 16 sequential pivots of block k it forms, for block k + 1, the products with every column that is
        // already final (p < r0); after the barrier only the 16 columns of block k are missing (four MFMAs per tile, above)
        const int r1 = r0 + 16;
        const double* pb = W + (size_t)(r1 + (lane & 15)) * ldw + (lane >> 4);
        for (int t = 0; t < ntile - 1; ++t) {
          solve_d4 acc = {0.0, 0.0, 0.0, 0.0};
          const double* pa = W + (size_t)(r1 + 16 * t + (lane & 15)) * ldw + (lane >> 4);
          for (int p = 0; p < r0; p += 16) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = pa[p + 4 * u]; bv[u] = pb[p + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
          }
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) pnext[(16 * t + 4 * reg + (lane >> 4)) * 17 + (lane & 15)] = acc[reg];
        }
      }
      if (wave == 0 || 16 + 48 * wave < 16 * ntile || (LDSW && wave == NW - 1)) {  // wavefronts whose rows are all past the end sit this panel out
        const bool rv = q < 16 * ntile;
        double r[16];
        {
          const double* wsrc = W + (size_t)(r0 + (rv ? q : 0)) * ldw + r0;
#pragma unroll
          for (int c = 0; c < 16; ++c) r[c] = wsrc[c];
          if (k > 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] -= panel[(rv ? q : 0) * 17 + c];
          }
          if (idrow) {
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] = (c == lane - 48) ? 1.0 : 0.0;
          }
        }
        double myinv = 1.0;
        // (round 6) the block that holds the right-hand-side row is the last one: from row n on its rows are the right-hand side (not a pivot)
        // and identity padding -- elimination steps that multiply by 1 and subtract 0.  They are skipped: exact (bit-identical), and they are
        // the longest steps of the block (step j is j broadcasts + FMAs deep): 4 of 16 steps remain at 6 cameras with the intrinsics fixed
        // (n = 36), 8 of 16 at 2 and at 6 cameras with every parameter free (n = 24, 72)
        const int jmax = r0 + 16 <= n ? 16 : n - r0;
        // left-looking: column j of every row is finished at step j; L[r0+j][q'] (q' < j) is lane j's finished r[q'].
        // A non-positive pivot turns into NaN / inf here and surfaces as a non-finite step below.
        // (Round 3, measured and withdrawn: lane j carrying its pivot-to-be a_jj - sum L[j][q]^2 as a running accumulator, so that
        // only ONE broadcast sits on the dependent chain per pivot -- 23.2 k instead of 21.3 k cycles for this phase at 12C = 72:
        // the stream is bound by its 616 instructions per block, 240 of them v_readlane, not by the chain.)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (j < jmax) {   // (wave-uniform; a `break` here would keep the loop from unrolling and r[] from living in registers)
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int qq = 0; qq < j; ++qq) {
              const double ljq = lane_bcast(r[qq], j);
              if (qq & 1) s1 = fma(r[qq], ljq, s1); else s0 = fma(r[qq], ljq, s0);
            }
            r[j] -= s0 + s1;
            const double pj = lane_bcast(r[j], j);
            const double inv = rsqrt_cubic(pj);
            r[j] *= inv;
            myinv = lane == j ? inv : myinv;
          }
        }
        if (rv && (lane >= 16 || wave == 0)) {
          double* wr = W + (size_t)(r0 + q) * ldw + r0;
#pragma unroll
          for (int c = 0; c < 16; ++c) wr[c] = r[c];
        }
        if (wave == 0 && lane < 16) invd[r0 + lane] = myinv;
        if (idrow) {
          double* li = linv + (k * 16 + (lane - 48)) * 17;
#pragma unroll
          for (int c = 0; c < 16; ++c) li[c] = r[c];
        }
      }
      __syncthreads();
      LAP(1);
    }

    STAMP(2);
    // ---- backward sweep  L^T d = y,  y = row n of the factor
    for (int j = tid; j < npad; j += NTHREADS) yv[j] = j < n ? W[(size_t)n * ldw + j] : 0.0;
    __syncthreads();
    {
      // d_k = L_kk^-T y_k: row i of the inverse transpose (zero left of the diagonal) times the block's right-hand side; y
      // beyond row n is zero, so the right-hand-side row and the padding drop out.  Then y_j -= sum_t L[r0+t][j] d[r0+t].
      for (int k = nblk - 1; k >= 0; --k) {
        const int r0 = 16 * k;
        double wcur[16];  // wcur[t] = L[r0 + t][j] for this thread's column j = tid: in flight during the dot products
#pragma unroll
        for (int t = 0; t < 16; ++t) wcur[t] = W[(size_t)(r0 + t) * ldw + min(tid, npad - 1)];
        if (tid < 16) {
          const double* li = linv + (k * 16 + tid) * 17;
          double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 0; c < 16; ++c) s[c & 3] = fma(li[c], yv[r0 + c], s[c & 3]);
          dv[r0 + tid] = (r0 + tid < n) ? (s[0] + s[1]) + (s[2] + s[3]) : 0.0;
        }
        __syncthreads();
        for (int j = tid; j < r0; j += NTHREADS) {
          double s0 = yv[j], s1 = 0.0;
          if (j == tid) {
#pragma unroll
            for (int t = 0; t < 16; t += 2) { s0 = fma(-wcur[t], dv[r0 + t], s0); s1 = fma(-wcur[t + 1], dv[r0 + t + 1], s1); }
          } else {
            const double* wc = W + (size_t)r0 * ldw + j;
#pragma unroll
            for (int t = 0; t < 16; t += 2) { s0 = fma(-wc[(size_t)t * ldw], dv[r0 + t], s0); s1 = fma(-wc[(size_t)(t + 1) * ldw], dv[r0 + t + 1], s1); }
          }
          yv[j] = s0 + s1;
        }
        __syncthreads();
      }
    }
    }
  }

  STAMP(3);
  // ---- step scalars, failure handling, state
  double sums[4] = {0.0, 0.0, 0.0, 0.0};  // pred_cam, |d_c|^2, |x_c|^2, non-finite entries
  if (mode == 0) {
    for (int i = tid; i < n; i += NTHREADS) {
      const double d = dv[i], xv = xc[a.cw == 12 ? i : 12 * (i / 6) + 6 + i % 6];  // (camera block 6 wide: the system's row i is parameter 6 + i % 6 of camera i / 6)
      a.dc[i] = d;
      sums[3] += (fabs(d) < 1e300) ? 0.0 : 1.0;
      sums[0] += d * (damp[i] * d - gc[i]);
      sums[1] += d * d;
      sums[2] += xv * xv;
    }
  }
  if (a.flag && mode == 0) {  // k_solve_backsub: the camera step is in memory -- the back-substitution workgroups may start on it
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flag, 4.0 * a.seq + 1.0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  block_reduce4<NTHREADS>(sums, false, s_red);
  if (tid == 0) {
    const bool failed = mode == 2 || sums[3] != 0.0 || !(fabs(sums[0]) < 1e300);
    lms[MCBA_LM_SOLVE_INFO] = mode == 2 ? 2.0 : failed ? 1.0 : 0.0;
    if (failed) {  // more damping; the next tick rebuilds the reduced system without a trial step
      const double lam = fmin(lambda * lms[2], a.lam_max);
      lms[1] = lam;
      lms[2] *= 2.0;
      lms[MCBA_LM_SKIP] = 1.0;
      if (lam >= a.lam_max) lms[MCBA_LM_DONE] = 3.0;
    } else {
      lms[MCBA_LM_SKIP] = 0.0;
      lms[MCBA_LM_PRED_CAM] = sums[0];
      lms[MCBA_LM_DCN2] = sums[1];
      lms[MCBA_LM_XCN2] = sums[2];
    }
  }
  __syncthreads();
  post_state(a, lst, true, a.flag != nullptr && mode == 0);
}

template <int NTHREADS, bool LDSW, int KS>
__global__ __launch_bounds__(NTHREADS) void k_solve_cam(SolveArgs a) {
  if constexpr (!LDSW) {
    if (blockIdx.x > 0) { rl_stager(a, (int)blockIdx.x - 1, (int)gridDim.x - 1); return; }
  }
  solve_cam_body<NTHREADS, LDSW, KS>(a);
}

// The reduced solve AND the back-substitution of the next trial step in one launch (single-GPU ticks, factor in LDS, <= 8
// cameras): workgroup 0 is k_solve_cam (its first four wavefronts), workgroups 1 .. Fpad / 64 are k_backsub -- they put
// their 39 MB of W blocks, the frame factors and the poses in flight at once, then wait for workgroup 0 to release the
// camera step.  The solve is one workgroup of latency; the memory system is idle meanwhile, so the back-substitution's
// whole load time disappears behind it and one launch is saved.  Workgroup 0 is dispatched first and waits for nobody.
struct BacksubArgs {
  Sel sl;
  const double *rec0, *rec1, *fbuf;
  double *x0, *x1, *bpart;
  int C, F, Fpad;
  BacksubWait wait;
};
template <int KS, int CW = 12>
__global__ __launch_bounds__(64 * kBacksubWaves) void k_solve_backsub(SolveArgs a, BacksubArgs b) {
  __shared__ double s_t[kBacksubWaves][6][64];
  __shared__ double s_mail[8 + 12 * 9];  // (the LDS-resident solve serves at most 9 cameras 12 wide, 18 cameras 6 wide: 112 rows)
  if (blockIdx.x == 0) {
    if (threadIdx.x < 256) solve_cam_body<256, true, KS>(a);
    return;
  }
  b.wait.mail = s_mail;
  backsub_body<DevStep, CW>(b.sl, b.rec0, b.rec1, b.fbuf, DevStep{a.dc}, b.x0, b.x1, b.bpart, b.C, b.F, b.Fpad, (int)blockIdx.x - 1, b.C < kBacksubWaves ? b.C : kBacksubWaves, s_t, &b.wait);
}

size_t solve_lds_bytes(int npad, int use_lds) {
  size_t d = (size_t)npad * 17 + 4 * (size_t)npad + 72 + MCBA_LMS;
  d += use_lds ? (size_t)npad * (npad + 1) + (size_t)(npad / 16) * 16 * 17 + (solve_lookahead(npad) ? (size_t)npad * 17 : 0) : (size_t)16 * (npad + 2) + 16 * 17 + 2 * 256;  // (+ the diagonal tile and two panel tiles of the right-looking variant)
  return d * sizeof(double);
}

// the factor fits LDS (and the register staging of the 256-thread variant covers it)
int solve_fits_lds(int npad, int lds_limit) { return npad <= 16 * kStageMax && solve_lds_bytes(npad, 1) <= (size_t)lds_limit - 10 * 1024; }  // (k_solve_backsub's static scratch rides along)

static const void* solve_kernel(int npad, int use_lds) {
  if (use_lds) return npad <= 80 ? reinterpret_cast<const void*>(&k_solve_cam<256, true, 5>) : reinterpret_cast<const void*>(&k_solve_cam<256, true, kStageMax>);
  return reinterpret_cast<const void*>(&k_solve_cam<512, false, 1>);
}

int solve_set_lds_limit(int npad, int use_lds) {
  size_t bytes = solve_lds_bytes(npad, use_lds);
  if (bytes <= 64 * 1024) return 0;
  return hipFuncSetAttribute(solve_kernel(npad, use_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : 1;
}

int solve_backsub_set_lds_limit(int npad, int cw) {
  const size_t bytes = solve_lds_bytes(npad, 1) + sizeof(double) * (kBacksubWaves * 6 * 64 + 8 + 12 * 9);
  if (bytes <= 64 * 1024) return 0;
  const void* k = cw == 6 ? (npad <= 80 ? reinterpret_cast<const void*>(&k_solve_backsub<5, 6>) : reinterpret_cast<const void*>(&k_solve_backsub<kStageMax, 6>))
                          : (npad <= 80 ? reinterpret_cast<const void*>(&k_solve_backsub<5>) : reinterpret_cast<const void*>(&k_solve_backsub<kStageMax>));
  return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds_bytes(npad, 1)) == hipSuccess ? 0 : 1;
}

void launch_solve_backsub(hipStream_t st, const SolveArgs& a, Sel sl, const double* rec0, const double* rec1, const double* fbuf, double* x0, double* x1, double* bpart, int C, int F, int Fpad,
                          const double* early_state, int max_polls, int spec, double* timeout_dev, double* timeout_host, int strict) {
  BacksubArgs b{sl, rec0, rec1, fbuf, x0, x1, bpart, C, F, Fpad, BacksubWait{early_state, a.flag, a.dc, nullptr, a.seq, max_polls, spec, a.lam_min, timeout_dev, timeout_host, a.dec_floor, strict}};
  const size_t lds = solve_lds_bytes(a.npad, 1);
  const dim3 grid(1 + Fpad / 64), block(64 * kBacksubWaves);
  if (a.cw == 6) {
    if (a.npad <= 80) hipLaunchKernelGGL((k_solve_backsub<5, 6>), grid, block, lds, st, a, b);
    else hipLaunchKernelGGL((k_solve_backsub<kStageMax, 6>), grid, block, lds, st, a, b);
  } else if (a.npad <= 80) hipLaunchKernelGGL((k_solve_backsub<5>), grid, block, lds, st, a, b);
  else hipLaunchKernelGGL((k_solve_backsub<kStageMax>), grid, block, lds, st, a, b);
}

void launch_solve_cam(hipStream_t st, const SolveArgs& a) {
  size_t lds = solve_lds_bytes(a.npad, a.use_lds);
  if (a.use_lds) {
    if (a.npad <= 80) hipLaunchKernelGGL((k_solve_cam<256, true, 5>), dim3(1), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_solve_cam<256, true, kStageMax>), dim3(1), dim3(256), lds, st, a);
    return;
  }
  int ns = rl_stagers(a.npad);
  SolveArgs b = a;
  if (const char* e = getenv("MCBA_SOLVE_STAGERS")) {  // 0: workgroup 0 brings the tiles in itself; -1 (tests): stagers are launched but do nothing, so workgroup 0's bounded wait runs out; -2 (tests): late stagers
    const int v = atoi(e);
    if (v == -2) b.stage_tag += 0.5;            // (tests: stagers that show up ~0.2 s late -- after workgroup 0 has taken their shares)
    else if (v < 0) b.stage_tag = -b.stage_tag;
    else ns = std::min(kRlMaxStagers, v);
  }
  hipLaunchKernelGGL((k_solve_cam<512, false, 1>), dim3(1 + ns), dim3(512), lds, st, b);  // workgroup 0 solves, the others bring the system's tiles in
}

}  // namespace mcba
