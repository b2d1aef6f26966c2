#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double r0 = __builtin_amdgcn_rcp(a);
  double e = fma(-a, r0, 1.0);
  double r1 = fma(r0, e, r0);
  e = fma(-a, r1, 1.0);
  double r2 = fma(r1, e, r1);
  double y0 = __builtin_amdgcn_rsq(a);
  double f = fma(-a * y0, y0, 1.0);
  double y1 = fma(0.5 * y0, f, y0);
  f = fma(-a * y1, y1, 1.0);
  double y2 = fma(0.5 * y1, f, y1);
  // one third-order step: y (1 + e/2 + 3 e^2 / 8)
  f = fma(-a * y0, y0, 1.0);
  double q = f * fma(0.375, f, 0.5);
  double y3 = fma(y0, q, y0);
  out[6 * i] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2; out[6 * i + 3] = y0; out[6 * i + 4] = y1; out[6 * i + 5] = y2 ;
  out[6 * n + i] = y3;
}
int main() {
  const int n = 1 << 20;
  double* hx = (double*)malloc(n * 8); double* ho = (double*)malloc(n * 7 * 8);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX) * 20.0 - 10.0);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 7 * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, n * 7 * 8, hipMemcpyDeviceToHost);
  double m[7] = {0};
  for (int i = 0; i < n; ++i) {
    long double rc = 1.0L / hx[i], rs = 1.0L / sqrtl(hx[i]);
    for (int k = 0; k < 3; ++k) m[k] = fmax(m[k], fabs((double)((ho[6 * i + k] - rc) / rc)));
    for (int k = 3; k < 6; ++k) m[k] = fmax(m[k], fabs((double)((ho[6 * i + k] - rs) / rs)));
    m[6] = fmax(m[6], fabs((double)((ho[6 * n + i] - rs) / rs)));
  }
  printf("rcp: seed %.3e  1 step %.3e  2 steps %.3e | rsq: seed %.3e  1 step %.3e  2 steps %.3e  cubic %.3e  (eps %.3e)\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6], 2.220446049250313e-16);
}
