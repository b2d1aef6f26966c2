// 16 x 16 Cholesky in one wavefront, row n in lanes with (lane & 15) == n: (a) v_readlane left-looking (what k_solve_cam does),
// (b) right-looking with v_fmac_f64_dpp row_newbcast.  Prints cycles per block and the max difference to a host factorisation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
  b.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
  return b.d;
}
__device__ __forceinline__ double rsqrt_cubic(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double e = fma(-a * y, y, 1.0);
  double q = e * fma(0.375, e, 0.5);
  return fma(y, q, y);
}
template <int J>
__device__ __forceinline__ double bc(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
// acc -= bcast_C(x) * x   (x was just written by a VALU instruction when FIRST: two wait states before a DPP read)
template <int C, bool FIRST>
__device__ __forceinline__ void fnmac_bc(double& acc, double x) {
  if constexpr (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "n"(C));
  else asm volatile("v_fmac_f64_dpp %0, -%1, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "n"(C));
}
template <int J, int C>
__device__ __forceinline__ void upd(double (&d)[16]) {
  if constexpr (C < 16) {
    fnmac_bc<C, C == J + 1>(d[C], d[J]);
    upd<J, C + 1>(d);
  }
}
template <int J>
__device__ __forceinline__ void step(double (&d)[16], double (&inv)[16]) {
  if constexpr (J < 16) {
    double p = bc<J>(d[J]);
    double y = rsqrt_cubic(p);
    d[J] *= y;
    inv[J] = y;
    upd<J, J + 1>(d);
    step<J + 1>(d, inv);
  }
}
template <int MODE>
__global__ void k(const double* A, double* L, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63, n = lane & 15;
  double d0[16];
  for (int c = 0; c < 16; ++c) d0[c] = A[n * 16 + c];
  double d[16], inv[16];
  long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = d0[c] + 1e-300 * rep;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int qq = 0; qq < j; ++qq) {
          const double ljq = lane_bcast(d[qq], j);
          if (qq & 1) s1 = fma(d[qq], ljq, s1); else s0 = fma(d[qq], ljq, s0);
        }
        d[j] -= s0 + s1;
        double pj = lane_bcast(d[j], j);
        const double iv = rsqrt_cubic(pj);
        d[j] *= iv;
        inv[j] = iv;
      }
    } else {
      step<0>(d, inv);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) d0[c] += 1e-300 * d[c];
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (lane < 16) for (int c = 0; c < 16; ++c) L[n * 16 + c] = d[c];
}
int main() {
  std::vector<double> A(256), Lh(256, 0.0), Lg(256);
  srand(1);
  std::vector<double> B(256);
  for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 4.0 : 0.0; for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k]; A[i * 16 + j] = s; }
  for (int j = 0; j < 16; ++j) {
    double s = A[j * 16 + j];
    for (int q = 0; q < j; ++q) s -= Lh[j * 16 + q] * Lh[j * 16 + q];
    Lh[j * 16 + j] = sqrt(s);
    for (int i = j + 1; i < 16; ++i) { double t = A[i * 16 + j]; for (int q = 0; q < j; ++q) t -= Lh[i * 16 + q] * Lh[j * 16 + q]; Lh[i * 16 + j] = t / Lh[j * 16 + j]; }
  }
  double *dA, *dL; long long* dc;
  hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    for (int reps : {1, 101}) {
      if (mode == 0) k<0><<<1, 64>>>(dA, dL, dc, reps); else k<1><<<1, 64>>>(dA, dL, dc, reps);
      hipDeviceSynchronize();
      long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(Lg.data(), dL, 2048, hipMemcpyDeviceToHost);
      double e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) e = fmax(e, fabs(Lg[i * 16 + j] - Lh[i * 16 + j]));
      printf("mode %d reps %d cycles %lld per block %.0f maxdiff %.3e\n", mode, reps, c, (double)c / reps, e);
    }
  }
  return 0;
}
