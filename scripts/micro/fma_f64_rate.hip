// v_fma_f64 issue rate on gfx950 with ONE wavefront per SIMD: cycles per instruction for 32 independent accumulators.
// hipcc -O3 --offload-arch=gfx950 -o fma_f64_rate fma_f64_rate.hip && ./fma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(double* out, long long* cyc, int iters, double a, double b) {
  double acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x + i;
  double x = a + threadIdx.x * 1e-3, y = b;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = fma(acc[i], x, y);
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = fma(acc[i], y, x);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out = nullptr; long long* cyc = nullptr;
  if (hipMalloc(&out, (size_t)256 * 1024 * 8) != hipSuccess || hipMalloc(&cyc, 256 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const int iters = 2000;
  for (int waves : {4, 8, 16}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (waves == 4) k<4><<<256, 256>>>(out, cyc, iters, 1.0000001, 1e-9);
      else if (waves == 8) k<8><<<256, 512>>>(out, cyc, iters, 1.0000001, 1e-9);
      else k<16><<<256, 1024>>>(out, cyc, iters, 1.0000001, 1e-9);
      (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = 64.0 * iters;
    printf("%2d waves/CU: %.2f clock64 ticks per v_fma_f64 (wave 0); kernel %.3f ms -> %.1f TFLOP/s\n", waves, c / n, ms, 256.0 * waves * 64 * n * 2 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
