// How a 52 MB host array (pageable, as numpy hands it over) gets to the GPU fastest while kernels consume it chunk by chunk
// (development aid behind mcba_upload_scored: the pre-filter's scoring under the PCIe transfer).
//   A  one hipMemcpyAsync + sync                      B  K chunks on a copy stream, a consumer kernel per chunk on a second stream
//   C  the same on ONE stream                          D  hipHostRegister + one copy + unregister
//   E  staged through a pinned ring by the host (1 thread memcpy)
//   G  small D2H copies: 480 KB flat, 10 000 x 48 B strided (hipMemcpy2DAsync), 1 KB          H  empty-launch + sync round trip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
__global__ void k_consume(const double2* __restrict__ a, double* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 123.456) o[0] = s;
}
__global__ void k_empty() {}
int main() {
  const size_t C = 6, F = 10000, N = 54, bytes = C * F * N * 16;
  double* src = static_cast<double*>(malloc(bytes));
  for (size_t i = 0; i < bytes / 8; ++i) src[i] = 1e-3 * (double)(i & 1023);
  double *dev, *sink, *small;
  hipMalloc(&dev, bytes); hipMalloc(&sink, 64); hipMalloc(&small, 1 << 20);
  hipStream_t cs, ks;
  hipStreamCreateWithFlags(&cs, hipStreamNonBlocking); hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
  std::vector<hipEvent_t> ev(64);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  {  // A
    std::vector<double> t;
    for (int r = 0; r < 7; ++r) { auto t0 = clk::now(); hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, cs); double tc = ms_since(t0); hipStreamSynchronize(cs); t.push_back(ms_since(t0)); if (r == 6) printf("A one copy: call returns after %.3f ms\n", tc); }
    printf("A one hipMemcpyAsync %.1f MB pageable + sync: median %.3f ms (%.1f GB/s), min %.3f\n", bytes / 1e6, med(t), bytes / med(t) / 1e6, *std::min_element(t.begin(), t.end()));
  }
  for (int one_stream = 0; one_stream < 2; ++one_stream)
    for (int K : {6, 12, 24, 48}) {
      std::vector<double> t, tcall, ttail;
      for (int r = 0; r < 6; ++r) {
        const size_t cb = bytes / K;
        auto t0 = clk::now();
        double call = 0, last_ret = 0;
        for (int k = 0; k < K; ++k) {
          auto t1 = clk::now();
          hipMemcpyAsync(reinterpret_cast<char*>(dev) + k * cb, reinterpret_cast<char*>(src) + k * cb, cb, hipMemcpyHostToDevice, cs);
          call += ms_since(t1);
          hipStream_t kk = one_stream ? cs : ks;
          if (!one_stream) { hipEventRecord(ev[k], cs); hipStreamWaitEvent(ks, ev[k], 0); }
          k_consume<<<256, 256, 0, kk>>>(reinterpret_cast<const double2*>(reinterpret_cast<char*>(dev) + k * cb), sink, cb / 16);
          last_ret = ms_since(t0);
        }
        hipStreamSynchronize(cs); hipStreamSynchronize(ks);
        t.push_back(ms_since(t0)); tcall.push_back(call); ttail.push_back(ms_since(t0) - last_ret);
      }
      printf("%s K=%2d chunks of %.2f MB + consumer kernel each: total median %.3f ms, time inside the copy calls %.3f, tail after the last launch %.3f\n", one_stream ? "C one stream " : "B two streams", K, bytes / K / 1e6,
             med(t), med(tcall), med(ttail));
    }
  {  // D
    std::vector<double> t, tr, tu;
    for (int r = 0; r < 5; ++r) {
      auto t0 = clk::now();
      hipError_t e = hipHostRegister(src, bytes, hipHostRegisterDefault);
      double t_reg = ms_since(t0);
      hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, cs); hipStreamSynchronize(cs);
      double t_copy = ms_since(t0);
      if (e == hipSuccess) hipHostUnregister(src);
      t.push_back(ms_since(t0)); tr.push_back(t_reg); tu.push_back(ms_since(t0) - t_copy);
      if (e != hipSuccess) printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
    }
    printf("D register %.3f + copy + unregister %.3f: total median %.3f ms\n", med(tr), med(tu), med(t));
  }
  {  // E
    const size_t ring_chunk = 4 << 20;
    char* ring;
    hipHostMalloc(reinterpret_cast<void**>(&ring), 4 * ring_chunk, hipHostMallocDefault);
    std::vector<double> t;
    for (int r = 0; r < 5; ++r) {
      auto t0 = clk::now();
      size_t off = 0; int k = 0;
      while (off < bytes) {
        const size_t cb = std::min(ring_chunk, bytes - off);
        if (k >= 4) hipEventSynchronize(ev[k % 4]);
        memcpy(ring + (k % 4) * ring_chunk, reinterpret_cast<char*>(src) + off, cb);
        hipMemcpyAsync(reinterpret_cast<char*>(dev) + off, ring + (k % 4) * ring_chunk, cb, hipMemcpyHostToDevice, cs);
        hipEventRecord(ev[k % 4], cs);
        off += cb; ++k;
      }
      hipStreamSynchronize(cs);
      t.push_back(ms_since(t0));
    }
    printf("E staged through a 4 x 4 MiB pinned ring (one host thread): median %.3f ms\n", med(t));
    hipHostFree(ring);
  }
  {  // G
    double* hs = static_cast<double*>(malloc(1 << 20));
    double* hp; hipHostMalloc(reinterpret_cast<void**>(&hp), 1 << 20, hipHostMallocDefault);
    for (int what = 0; what < 6; ++what) {
      std::vector<double> t;
      for (int r = 0; r < 9; ++r) {
        auto t0 = clk::now();
        if (what == 0) hipMemcpyAsync(hs, small, 480000, hipMemcpyDeviceToHost, cs);
        if (what == 1) hipMemcpy2DAsync(hs, 48, small, 320, 48, 3000, hipMemcpyDeviceToHost, cs);
        if (what == 2) hipMemcpyAsync(hs, small, 1024, hipMemcpyDeviceToHost, cs);
        if (what == 3) hipMemcpyAsync(hp, small, 480000, hipMemcpyDeviceToHost, cs);
        if (what == 4) hipMemcpyAsync(hp, small, 1024, hipMemcpyDeviceToHost, cs);
        if (what == 5) hipMemcpyAsync(small, hs, 480000, hipMemcpyHostToDevice, cs);
        hipStreamSynchronize(cs);
        t.push_back(ms_since(t0));
      }
      const char* names[] = {"D2H 480 KB -> pageable", "D2H 2D 3000 rows x 48 B (pitch 320) -> pageable", "D2H 1 KB -> pageable", "D2H 480 KB -> pinned", "D2H 1 KB -> pinned", "H2D 480 KB pageable"};
      printf("G %-50s median %.1f us, min %.1f\n", names[what], 1e3 * med(t), 1e3 * *std::min_element(t.begin(), t.end()));
    }
  }
  {  // H
    std::vector<double> t, t2, t3;
    for (int r = 0; r < 20; ++r) {
      auto t0 = clk::now(); k_empty<<<1, 64, 0, cs>>>(); hipStreamSynchronize(cs); t.push_back(ms_since(t0));
      t0 = clk::now(); hipStreamSynchronize(cs); t2.push_back(ms_since(t0));
      t0 = clk::now(); for (int i = 0; i < 10; ++i) k_empty<<<1, 64, 0, cs>>>(); hipStreamSynchronize(cs); t3.push_back(ms_since(t0));
    }
    printf("H empty launch + sync %.1f us; sync of an idle stream %.1f us; 10 empty launches + sync %.1f us\n", 1e3 * med(t), 1e3 * med(t2), 1e3 * med(t3));
  }
  return 0;
}
