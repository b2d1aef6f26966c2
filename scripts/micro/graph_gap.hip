// Does a HIP graph shorten the boundary between dependent kernels?  K dependent small kernels (a) launched into a stream, (b) captured
// once into a graph and replayed (development aid: the LM tick is four dependent kernels; at shapes below one round of the chip the three
// boundaries are ~6 of its 44 us).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
using clk = std::chrono::steady_clock;
__global__ void k_touch(double* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
int main() {
  double* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto us = [](clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
  for (int nb : {1, 96, 256}) {
    auto run_stream = [&](int K) { auto t0 = clk::now(); for (int i = 0; i < K; ++i) k_touch<<<nb, 256, 0, s>>>(buf, nb * 256); hipStreamSynchronize(s); return us(t0); };
    std::vector<double> a, b;
    for (int r = 0; r < 7; ++r) { const double t1 = run_stream(40), t2 = run_stream(440); a.push_back((t2 - t1) / 400.0); }
    // graph of 4 dependent kernels (one "tick"), replayed
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 4; ++i) k_touch<<<nb, 256, 0, s>>>(buf, nb * 256);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    auto run_graph = [&](int K) { auto t0 = clk::now(); for (int i = 0; i < K; ++i) hipGraphLaunch(ge, s); hipStreamSynchronize(s); return us(t0); };
    for (int r = 0; r < 7; ++r) { const double t1 = run_graph(10), t2 = run_graph(110); b.push_back((t2 - t1) / 400.0); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%3d workgroups: dependent kernels in a stream %.2f us each; as graphs of four, replayed: %.2f us per kernel\n", nb, a[3], b[3]);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
