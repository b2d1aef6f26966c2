// obs_lead with a planar board: general form vs planar form on the GPU, bit for bit (development check).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../multicam-calibration_amd/csrc/mcba_math.h"
using namespace mcba;
__global__ void k(const double* in, int* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PairConst pc;
  for (int j = 0; j < 9; ++j) pc.Rcf[j] = in[16 * i + j];
  for (int j = 0; j < 3; ++j) pc.tcf[j] = in[16 * i + 9 + j];
  double X[3] = {in[16 * i + 12], in[16 * i + 13], 0.0};
  ObsLead a, b;
  obs_lead<true, false>(pc, X, a, true);
  obs_lead<true, true>(pc, X, b, true);
  bool same = a.xr[0] == b.xr[0] && a.xr[1] == b.xr[1] && a.xr[2] == b.xr[2] && a.iz == b.iz && a.a == b.a && a.b == b.b;
  if (!same) atomicAdd(bad, 1);
}
int main() {
  const int n = 1 << 20;
  double* h = (double*)malloc(sizeof(double) * 16 * n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < 9; ++j) h[16 * i + j] = (rand() / (double)RAND_MAX - 0.5) * 2;
    for (int j = 0; j < 3; ++j) h[16 * i + 9 + j] = (rand() / (double)RAND_MAX) * 500 + 100;
    h[16 * i + 12] = (rand() / (double)RAND_MAX) * 100; h[16 * i + 13] = (rand() / (double)RAND_MAX) * 100;
  }
  double* d; int* bad; int hb = 0;
  hipMalloc(&d, sizeof(double) * 16 * n); hipMalloc(&bad, 4);
  hipMemcpy(d, h, sizeof(double) * 16 * n, hipMemcpyHostToDevice); hipMemcpy(bad, &hb, 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, bad, n);
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("mismatches on the GPU: %d of %d\n", hb, n);
  return 0;
}
