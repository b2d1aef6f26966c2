// What a grid-wide phase boundary costs INSIDE a persistent kernel on this part, against what a kernel boundary costs in a stream
// (development aid behind the decision on a one-launch LM tick for shapes below one round of the chip: VERDICT r4 task 2).
//   barrier A  flat: one counter, every workgroup arrives with an agent-scope atomic and polls it
//   barrier B  two levels: 8 group counters (workgroup % 8 ~ XCD), the last of a group arrives at the top counter, the last of all
//              publishes the epoch in 8 flags (one per group) that the workgroups of the group poll
//   barrier C  as B, but every workgroup polls ONE flag
//   launches   K dependent empty kernels in one stream (the boundary the persistent kernel would replace)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
using clk = std::chrono::steady_clock;
struct Bar { unsigned flat; unsigned pad0[31]; unsigned sub[8][32]; unsigned top; unsigned pad1[31]; unsigned flag[8][32]; unsigned one; };
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int MODE>
__global__ void k_bar(Bar* b, int iters, double* sink) {
  const unsigned nb = gridDim.x, g = blockIdx.x % 8, ng = (nb + 7 - g) / 8;
  double acc = threadIdx.x;
  for (int it = 1; it <= iters; ++it) {
    acc = acc * 1.0000001 + 1.0;   // (a little work between the barriers)
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (MODE == 0) {
        atomicAdd(&b->flat, 1u);
        for (int spin = 0; spin < 4000000 && ld(&b->flat) < (unsigned)it * nb; ++spin) __builtin_amdgcn_s_sleep(1);   // (bounded: every wave reaches the end)
      } else {
        const unsigned a = atomicAdd(&b->sub[g][0], 1u);
        if (a == (unsigned)it * ng - 1) {
          const unsigned t = atomicAdd(&b->top, 1u);
          if (t == (unsigned)it * 8 - 1) {
            if (MODE == 1) for (int k = 0; k < 8; ++k) st(&b->flag[k][0], (unsigned)it);
            else st(&b->one, (unsigned)it);
          }
        }
        const unsigned* f = MODE == 1 ? &b->flag[g][0] : &b->one;
        for (int spin = 0; spin < 4000000 && ld(f) < (unsigned)it; ++spin) __builtin_amdgcn_s_sleep(1);
      }
      __threadfence();
    }
    __syncthreads();
  }
  if (acc == 1.2345) sink[0] = acc;
}
__global__ void k_empty(double* sink) { if (threadIdx.x == 1000) sink[0] = 1.0; }
int main() {
  Bar* b; double* sink;
  hipMalloc(&b, sizeof(Bar)); hipMalloc(&sink, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto run = [&](int mode, int nb, int iters) {
    std::vector<double> t;
    for (int r = 0; r < 5; ++r) {
      hipMemsetAsync(b, 0, sizeof(Bar), s); hipStreamSynchronize(s);
      auto t0 = clk::now();
      if (mode == 0) k_bar<0><<<nb, 256, 0, s>>>(b, iters, sink);
      if (mode == 1) k_bar<1><<<nb, 256, 0, s>>>(b, iters, sink);
      if (mode == 2) k_bar<2><<<nb, 256, 0, s>>>(b, iters, sink);
      hipStreamSynchronize(s);
      t.push_back(std::chrono::duration<double, std::micro>(clk::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[2];
  };
  for (int nb : {8, 32, 96, 204, 256}) {
    for (int mode = 0; mode < 3; ++mode) {
      const double t1 = run(mode, nb, 20), t2 = run(mode, nb, 220);
      printf("%3d workgroups, barrier %c: %.2f us per barrier\n", nb, "ABC"[mode], (t2 - t1) / 200.0);
    }
  }
  for (int nb : {1, 96, 256}) {
    std::vector<double> t;
    for (int r = 0; r < 7; ++r) {
      hipStreamSynchronize(s);
      auto t0 = clk::now();
      for (int i = 0; i < 20; ++i) k_empty<<<nb, 256, 0, s>>>(sink);
      hipStreamSynchronize(s);
      const double a = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
      t0 = clk::now();
      for (int i = 0; i < 220; ++i) k_empty<<<nb, 256, 0, s>>>(sink);
      hipStreamSynchronize(s);
      t.push_back((std::chrono::duration<double, std::micro>(clk::now() - t0).count() - a) / 200.0);
    }
    std::sort(t.begin(), t.end());
    printf("dependent empty kernels of %3d workgroups in one stream: %.2f us each\n", nb, t[3]);
  }
  return 0;
}
