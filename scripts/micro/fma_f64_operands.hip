// v_fma_f64 issue rate on gfx950, one wavefront per SIMD, by OPERAND PATTERN (round 4): k_gram's point loop runs at ~6.5 shader
// cycles per FP64 instruction although independent v_fma_f64 on two shared operands issue every 4.8 (fma_f64_rate.hip).
//   MODE 0  acc[i] = fma(acc[i], x, y)          one accumulator + two operands shared by every instruction (the old micro-benchmark)
//   MODE 1  acc[6 i + j] += u[i] * v[j]         the Gram update: three distinct register pairs per instruction, 36 accumulators
//   MODE 2  acc[i] += u[i] * s                  one operand in SGPRs (wave-uniform)
//   MODE 3  as 1, with a v_mul_f64 per six fmas (uw[i] = w * u[i]), the accumulate block's real mix
// hipcc -O3 --offload-arch=gfx950 -o fma_f64_operands fma_f64_operands.hip && ./fma_f64_operands
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, double a, double b) {
  double acc[36], u[6], v[6];
  for (int i = 0; i < 36; ++i) acc[i] = threadIdx.x + i;
  for (int i = 0; i < 6; ++i) { u[i] = a + 1e-3 * (threadIdx.x + i); v[i] = b + 1e-4 * (threadIdx.x - i); }
  double x = a + threadIdx.x * 1e-3, y = b, w = 1.0 + 1e-9 * threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 36; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[6 * i + j]) : "v"(u[i]), "v"(v[j]));
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 36; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(u[i % 6]), "s"(b));
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double uw;
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(uw) : "v"(u[i]), "v"(w));
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[6 * i + j]) : "v"(uw), "v"(v[j]));
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 36; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, double* out, long long* cyc, int per_iter) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, cyc, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %.2f clock64 ticks per FP64 instruction (wave 0 of workgroup 0), kernel %.3f ms\n", name, (double)c / ((double)iters * per_iter), ms);
}
int main() {
  double* out = nullptr; long long* cyc = nullptr;
  if (hipMalloc(&out, (size_t)256 * 256 * 8) != hipSuccess || hipMalloc(&cyc, 256 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<0>("0: acc = fma(acc, x, y), x y shared", out, cyc, 36);
  run<1>("1: acc[6i+j] += u[i] * v[j]", out, cyc, 36);
  run<2>("2: acc[i] += u[i] * sgpr", out, cyc, 36);
  run<3>("3: uw = w u[i]; acc[6i+j] += uw * v[j]", out, cyc, 42);
  return 0;
}
