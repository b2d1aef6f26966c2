// What a grid-wide phase boundary costs INSIDE a persistent kernel on this part, against what a kernel boundary costs in a stream
// (development aid behind the decision on a one-launch LM tick for shapes below one round of the chip: VERDICT r4 task 2).
//   barrier A  flat: one counter, every workgroup arrives with an agent-scope atomic and polls it
//   barrier B  two levels: 8 group counters (workgroup % 8 ~ XCD), the last of a group arrives at the top counter, the last of all
//              publishes the epoch in 8 flags (one per group) that the workgroups of the group poll
//   barrier C  as B, but every workgroup polls ONE flag
//   barrier D  no read-modify-write at all: every workgroup STORES the epoch in its own flag (stride 4 / 64 / 128 bytes), the first wavefront
//              of every workgroup polls all flags with vector loads (lane l reads flags l, l + 64, ...) until the smallest is the epoch
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
template <int STRIDE, int FENCE = 3>   // FENCE bit 0: release fence before the flag store, bit 1: acquire fence after the poll
__global__ void k_flags(unsigned* flags, int iters, double* sink) {
  const unsigned nb = gridDim.x;
  double acc = threadIdx.x;
  for (int it = 1; it <= iters; ++it) {
    acc = acc * 1.0000001 + 1.0;
    __syncthreads();
    if (threadIdx.x < 64) {
      if (threadIdx.x == 0) { if (FENCE & 1) __threadfence(); st(&flags[blockIdx.x * STRIDE], (unsigned)it); }
      for (int spin = 0; spin < 4000000; ++spin) {   // (bounded: every wave reaches the end)
        unsigned m = 0xffffffffu;
        for (unsigned k = threadIdx.x; k < nb; k += 64) m = min(m, ld(&flags[k * STRIDE]));
        if (__all(m >= (unsigned)it)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (FENCE & 2) __threadfence();
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
  unsigned* flags; hipMalloc(&flags, 256 * 32 * 4);
  auto runf = [&](int stride, int nb, int iters) {
    std::vector<double> t;
    for (int r = 0; r < 5; ++r) {
      hipMemsetAsync(flags, 0, 256 * 32 * 4, s); hipStreamSynchronize(s);
      auto t0 = clk::now();
      if (stride == 1) k_flags<1><<<nb, 256, 0, s>>>(flags, iters, sink);
      if (stride == 16) k_flags<16><<<nb, 256, 0, s>>>(flags, iters, sink);
      if (stride == 32) k_flags<32><<<nb, 256, 0, s>>>(flags, iters, sink);
      if (stride == 320) k_flags<32, 0><<<nb, 256, 0, s>>>(flags, iters, sink);
      if (stride == 321) k_flags<32, 1><<<nb, 256, 0, s>>>(flags, iters, sink);
      if (stride == 322) k_flags<32, 2><<<nb, 256, 0, s>>>(flags, iters, sink);
      hipStreamSynchronize(s);
      t.push_back(std::chrono::duration<double, std::micro>(clk::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[2];
  };
  for (int nb : {8, 32, 96, 204, 256})
    for (int stride : {1, 16, 32}) {
      const double t1 = runf(stride, nb, 20), t2 = runf(stride, nb, 220);
      printf("%3d workgroups, barrier D (flags %3d bytes apart): %.2f us per barrier\n", nb, stride * 4, (t2 - t1) / 200.0);
    }
  for (int nb : {8, 96, 204, 256})
    for (int f : {0, 1, 2}) {
      const double t1 = runf(320 + f, nb, 20), t2 = runf(320 + f, nb, 220);
      printf("%3d workgroups, barrier D (128 bytes apart) with %s: %.2f us per barrier\n", nb, f == 0 ? "NO fences (not a barrier for data: the floor of the flag traffic)" : f == 1 ? "the release fence only" : "the acquire fence only", (t2 - t1) / 200.0);
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
