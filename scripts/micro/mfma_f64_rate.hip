// v_mfma_f64_16x16x4_f64 issue rate on gfx950: cycles per MFMA for 1 wave/SIMD with 4 independent accumulators.
// hipcc -O3 --offload-arch=gfx950 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int iters, double a0, double b0) {
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = a0 + threadIdx.x, b = b0;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out = nullptr; long long* cyc = nullptr;
  hipMalloc(&out, (size_t)1024 * 1024 * 8);  // up to 1024 blocks x 1024 threads
  hipMalloc(&cyc, 1024 * 8);
  if (!out || !cyc) { printf("allocation failed\n"); return 1; }
  for (int blocks : {256, 512}) {
    for (int threads : {256, 512, 1024}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters = 2000;
      k<<<blocks, threads>>>(out, cyc, iters, 1.0, 1.0);
      hipEventRecord(e0);
      k<<<blocks, threads>>>(out, cyc, iters, 1.0, 1.0);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      double nm = 4.0 * iters;
      printf("blocks %4d threads %3d: %.1f clock64 ticks per MFMA (wave 0), kernel %.3f ms -> %.2f TFLOP/s\n", blocks, threads, c / nm, ms, blocks * (threads / 64) * nm * 2048 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
