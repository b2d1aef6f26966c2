// MFMA f64 16x16x4 fed from LDS the way k_syrk does it: 4 waves, PPW = 4 pairs, two operand sets, stride 98.
// hipcc -O3 --offload-arch=gfx950 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
  __shared__ double s_y[80 * 98];
  for (int i = threadIdx.x; i < 80 * 98; i += 256) s_y[i] = 1e-3 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int rowa[4], rowb[4];
  for (int k = 0; k < 4; ++k) {
    int ti = (wave + k) % 5, tj = (wave + 2 * k + 1) % 5;
    rowa[k] = (16 * ti + (lane & 15)) * 98 + (lane >> 4);
    rowb[k] = (16 * tj + (lane & 15)) * 98 + (lane >> 4);
  }
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int nks = 24;
    double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a0[k] = s_y[rowa[k]]; b0[k] = s_y[rowb[k]]; }
    for (int ks = 0; ks < nks; ks += 2) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] = s_y[rowa[k] + 4 * ks + 4]; b1[k] = s_y[rowb[k] + 4 * ks + 4]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] = a0[k] + 1.0; b1[k] = b0[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
      if (ks + 2 < nks) {
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { a0[k] = s_y[rowa[k] + 4 * ks + 8]; b0[k] = s_y[rowb[k] + 4 * ks + 8]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) { a0[k] = a1[k] + 1.0; b0[k] = b1[k]; }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k], b1[k], acc[k], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out = nullptr; long long* cyc = nullptr;
  if (hipMalloc(&out, (size_t)512 * 256 * 8) != hipSuccess || hipMalloc(&cyc, 512 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const int iters = 50;
  for (int mode = 0; mode < 2; ++mode) {
    for (int blocks : {256, 512}) {
      if (mode == 0) k<0><<<blocks, 256>>>(out, cyc, iters); else k<1><<<blocks, 256>>>(out, cyc, iters);
      if (mode == 0) k<0><<<blocks, 256>>>(out, cyc, iters); else k<1><<<blocks, 256>>>(out, cyc, iters);
      hipDeviceSynchronize();
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s blocks %d: %.1f clock64 ticks per MFMA\n", mode == 0 ? "operands from LDS " : "operands in regs  ", blocks, c / (double)(iters * 24 * 4));
    }
  }
  return 0;
}
