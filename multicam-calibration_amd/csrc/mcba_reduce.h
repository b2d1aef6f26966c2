// mcba_reduce.h -- body of k_reduce_system (the fixed-order second stage of the Schur reduction), shared by the stand-alone
// kernel (mcba_kernels.hip, 16 wavefronts per block) and by k_reduce_solve_backsub (mcba_solve.hip, 8 wavefronts per block).
//
// Sums the per-workgroup partials of k_syrk and the per-wavefront partials of k_gram into the reduce buffer (layout in
// include/mcba.h) with coalesced reads and a FIXED summation order (bit-reproducible; no FP64 atomics anywhere) -- the same
// order whatever the block size: a (tile pair, accumulator register) block is 16 SLICES (slice s sums the partials g = s,
// s + 16, ...), a wavefront takes 16 / NW of them, wavefront 0 adds the 16 slice sums in order.
//   blocks [0, 4 NP): one per (tile pair q, accumulator register reg) = 64 elements that are 512 contiguous bytes in every
//       k_syrk partial.  Elements that fall on a camera's diagonal block also need U_c (and column 12C needs g_c): those sums
//       over the frame blocks are contiguous runs of gpart[camera][k][frame block] -- one wavefront task each.  Off-diagonal
//       tiles are mirrored on write.
//   then wavefront tasks (NW per block): diag(U), g_c, the 16 scalars and -- speculative frame-sharded ticks -- the 8 trial
//       scalars.
#pragma once
#include "mcba_device.h"
#include "mcba_math.h"

namespace mcba {

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

template <int NW>
__device__ __forceinline__ void reduce_system_body(const ReduceArgs& r, int blk, double (*s_part)[64], double* s_u) {
  static_assert(NW == 16 || NW == 8, "16 slices over 16 or 8 wavefronts");
  constexpr int SP = 16 / NW;  // slices per wavefront
  const Sel& sl = r.sl;
  const double* __restrict__ spart = r.spart;
  const double* __restrict__ fpart = r.fpart;
  double* __restrict__ red = r.red;
  const int C = r.C, nfb = r.nfb, G = r.G, NP = r.NP, nfblocks = r.nfblocks, rank_slot = r.rank_slot;
  const int n = 12 * C;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t camstride = (size_t)MCBA_GP * nfb;
  if (blk < 4 * NP) {
    const int q = blk >> 2, reg = blk & 3;
    const int ti = r.tile_i[q], tj = r.tile_j[q];
    // ---- slice sums of the k_syrk partials.  They do not depend on the LM state: ALL of this wavefront's rows (G <= 512
    // workgroups -> at most 32 per slice) go in flight before anything waits for the state -- one memory round trip
    // where a loop over batches of eight paid one per batch, and the state read rides along.
    double pv[SP][32];
    {
      const double* p = spart + (size_t)(4 * q + reg) * G * 64 + lane;
#pragma unroll
      for (int h = 0; h < SP; ++h)
#pragma unroll
        for (int k = 0; k < 32; ++k) { const int g = wave + NW * h + 16 * k; pv[h][k] = g < G ? p[(size_t)g * 64] : 0.0; }
    }
    if (!sel_active(sl, false)) return;
    const double* __restrict__ gpart = sel_index(sl) ? r.gp1 : r.gp0;
    // ---- U_c / g_c terms of the elements that need them: element e = slice + 16 j, one wavefront task each; the
    // tasks' runs are loaded together as well
    const double* up[SP][4];
#pragma unroll
    for (int h = 0; h < SP; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = wave + NW * h + 16 * j;
        const int row = 16 * ti + (e >> 4) + 4 * reg, col = 16 * tj + (e & 15);
        up[h][j] = nullptr;
        if (row < n && col < n && row / 12 == col / 12) {
          int cam = row / 12, li = row - 12 * cam, lj = col - 12 * cam;
          int a = li <= lj ? li : lj, b = li <= lj ? lj : li;
          up[h][j] = gpart + cam * camstride + (size_t)tri12(a, b) * nfb;
        } else if (col == n && row < n) {
          int cam = row / 12, li = row - 12 * cam;
          up[h][j] = gpart + cam * camstride + (size_t)(78 + li) * nfb;
        }
      }
    double us[SP][4];
#pragma unroll
    for (int h = 0; h < SP; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) us[h][j] = 0.0;
    for (int base = 0; base < nfb; base += 256) {
      double w[SP][4][4];
#pragma unroll
      for (int h = 0; h < SP; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) { const int i = base + lane + 64 * k; w[h][j][k] = (up[h][j] && i < nfb) ? up[h][j][i] : 0.0; }
#pragma unroll
      for (int h = 0; h < SP; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) us[h][j] += (w[h][j][0] + w[h][j][1]) + (w[h][j][2] + w[h][j][3]);
    }
#pragma unroll
    for (int h = 0; h < SP; ++h) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double u = wave_sum63(us[h][j]);
        if (lane == 63) s_u[wave + NW * h + 16 * j] = u;
      }
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) s += pv[h][k];
      s_part[wave + NW * h][lane] = s;
    }
    __syncthreads();
    if (wave == 0) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += s_part[k][lane];
      const int row = 16 * ti + (lane >> 4) + 4 * reg, col = 16 * tj + (lane & 15);
      if (row < n && col < n) {
        double out = s_u[lane] - v;  // S0 = blockdiag(U) - sum Y Y^T
        red[(size_t)row * n + col] = out;
        if (ti != tj) red[(size_t)col * n + row] = out;
      } else if (col == n && row < n) {
        red[(size_t)n * n + row] = v - s_u[lane];  // rhs = sum Y z - g_c
      }
    }
    return;
  }
  // ---- diag(U), g_c, scalars: task id per wavefront
  if (!sel_active(sl, false)) return;
  const double* __restrict__ gpart = sel_index(sl) ? r.gp1 : r.gp0;
  const double* __restrict__ bpart = r.bpart;
  const int nbp = r.nbp;
  const int task = (blk - 4 * NP) * NW + wave;
  double* tail = red + (size_t)n * n + n;
  if (task < n) {  // diag U
    int cam = task / 12, l = task - 12 * cam;
    double v = run_sum(gpart + cam * camstride + (size_t)tri12(l, l) * nfb, nfb, lane);
    if (lane == 63) tail[task] = v;
  } else if (task < 2 * n) {  // g_c
    int jj = task - n, cam = jj / 12, l = jj - 12 * cam;
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
    // speculative (frame-sharded) ticks: the trial point's scalars [cost, pred_f, |d_f|^2, |x_f|^2, #pairs, 0, 0, 0] for the
    // all-reduce that follows -- the reduction is built from the trial linearisation, so its cost and pair count are the
    // trial point's; what used to be a launch of its own (k_sum_trial) is eight more wavefront tasks here
    const int jj = task - (2 * n + 16);
    double a = 0.0;
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
  }
}

}  // namespace mcba
