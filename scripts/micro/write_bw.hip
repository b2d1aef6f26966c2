// Streaming write / copy / read ceilings on this GPU (development aid: sets the roof k_jacobian is judged against).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(double2* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  double2 v = make_double2(1.0, 2.0);
  for (; i < n; i += st) o[i] = v;
}
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) o[i] = a[i];
}
__global__ void k_read(const double2* __restrict__ a, double* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 123.456) o[0] = s;
}
int main() {
  size_t bytes = (size_t)1 << 30, n = bytes / 16;
  double2 *a, *b; double* c;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, 8);
  hipMemset(a, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192, 65536}) {
    for (int w = 0; w < 3; ++w) {
      float best = 1e9;
      for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        if (w == 0) k_write<<<grid, 256>>>(b, n);
        else if (w == 1) k_copy<<<grid, 256>>>(a, b, n);
        else k_read<<<grid, 256>>>(a, c, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
      }
      double moved = (w == 1 ? 2.0 : 1.0) * bytes;
      printf("grid %6d %-5s %.3f ms  %.2f TB/s\n", grid, w == 0 ? "write" : w == 1 ? "copy" : "read", best, moved / best / 1e9);
    }
  }
  return 0;
}
