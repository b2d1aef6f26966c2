// mcba_device.h -- small device helpers shared by the kernel translation units (wave reductions on DPP moves, the
// selection of double-buffered operands from the device-resident LM state).
#pragma once
#include <hip/hip_runtime.h>
#include "mcba_kernels.h"
#include "mcba_lm.h"

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
// v_mov_b32_dpp with an UNDEFINED previous destination (mov_dpp): lanes the move does not write (rows masked out by
// the two row_bcast steps) hold garbage afterwards -- harmless, because only lane 63 of the final value is used -- and
// the compiler no longer has to zero the destination before every move (1 100 instructions per k_gram epilogue).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], CTRL, ROW_MASK, 0xF, true);
  b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], CTRL, ROW_MASK, 0xF, true);
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
// spec (frame-sharded ticks with ONE collective): the Schur reduction runs BEFORE the decision is known, on the
// prediction "trial step accepted, lambda' = max(lambda / 3, lambda_min)" -- except in a rebuild tick (state[SKIP] != 0),
// which reduces the current linearisation with the state's own damping.  k_solve_cam checks the prediction afterwards.
__device__ __forceinline__ bool sel_spec(const Sel& s) { return s.spec && s.lms[MCBA_LM_SKIP] == 0.0; }
__device__ __forceinline__ int sel_index(const Sel& s) { return s.lms ? ((static_cast<int>(s.lms[3]) ^ s.idx ^ (sel_spec(s) ? 1 : 0)) & 1) : s.idx; }
__device__ __forceinline__ double sel_lambda(const Sel& s) { return s.lms ? (sel_spec(s) ? lm_spec_lambda(s.lms[1], s.lam, s.dec) : s.lms[1]) : s.lam; }
// device-resident LM loop: after termination every kernel of a tick returns at once; a tick that follows a failed
// reduced solve skips its trial kernels (`trial` = true) and only rebuilds the system with the raised damping
__device__ __forceinline__ bool sel_active(const Sel& s, bool trial) {
  if (!s.lms) return true;
  if (s.lms[MCBA_LM_DONE] != 0.0) return false;
  return !(trial && s.lms[MCBA_LM_SKIP] != 0.0);
}

}  // namespace mcba
