// FP64 VALU on gfx950, one wavefront per SIMD: result latency of dependent chains vs issue rate of independent streams,
// the cost of v_accvgpr moves next to them, and the LDS ds_add_f64 rate.  Shader cycles per instruction (s_memtime).
// hipcc -O3 --offload-arch=gfx950 -o f64_latency f64_latency.hip && ./f64_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
// MODE 0: dependent v_fma_f64 chain            1: dependent v_mul_f64 chain        2: dependent v_add_f64 chain
// MODE 3: 16 independent v_fma_f64             4: 16 independent fma + 16 v_accvgpr_write/read pairs interleaved
// MODE 5: dependent v_rcp_f64                  6: 8 independent fma, then each result consumed 8 instructions later
// MODE 7: 16 independent ds_add_f64 (no return) per iteration        8: fma stream + 4 ds_add_f64 per 16 fma
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, double a, double b) {
  __shared__ double lds[16][256];
  double acc[16];
  for (int i = 0; i < 16; ++i) { acc[i] = threadIdx.x + i; lds[i][threadIdx.x] = 0.0; }
  double x = a + threadIdx.x * 1e-3, y = b;
  double* lp = &lds[0][threadIdx.x];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[0]) : "v"(x), "v"(y));) }
    if constexpr (MODE == 1) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[0]) : "v"(x));) }
    if constexpr (MODE == 2) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[0]) : "v"(y));) }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    if constexpr (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        int t;
        asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %0, a1" : "=v"(t) : "v"(i) : "a0");
      }
    }
    if constexpr (MODE == 5) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(acc[0]));) }
    if constexpr (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(acc[8 + i]), "v"(x), "v"(y));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(acc[8 + i]) : "v"(acc[i]), "v"(x), "v"(y));
    }
    if constexpr (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"((unsigned)(size_t)lp), "v"(x), "n"(i * 2048) : "memory");
    }
    if constexpr (MODE == 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        if ((i & 3) == 0) asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"((unsigned)(size_t)lp), "v"(x), "n"(i * 2048) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + lds[i][threadIdx.x];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* what, double* out, long long* cyc, int per_iter) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, cyc, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.2f cycles per instruction  (kernel %.3f ms, %.2f GHz)\n", what, (double)c / ((double)iters * per_iter), ms, c / (ms * 1e6));
}
int main() {
  double* out = nullptr; long long* cyc = nullptr;
  if (hipMalloc(&out, (size_t)256 * 256 * 8) != hipSuccess || hipMalloc(&cyc, 256 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<0>("dependent v_fma_f64 chain", out, cyc, 16);
  run<1>("dependent v_mul_f64 chain", out, cyc, 16);
  run<2>("dependent v_add_f64 chain", out, cyc, 16);
  run<5>("dependent v_rcp_f64 chain", out, cyc, 16);
  run<3>("16 independent v_fma_f64", out, cyc, 16);
  run<6>("v_fma_f64, consumer 8 instructions after producer", out, cyc, 16);
  run<4>("16 independent v_fma_f64 + 32 accvgpr moves (per fma)", out, cyc, 16);
  run<7>("ds_add_f64, 4 waves per CU (per ds_add)", out, cyc, 16);
  run<8>("16 v_fma_f64 + 4 ds_add_f64 (per fma)", out, cyc, 16);
  return 0;
}
