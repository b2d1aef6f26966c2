// The 16-tile k_syrk inner loop in isolation (24 cameras: 19 tile rows, row stride 26 doubles, 6 K-steps per stage, 16 tile
// pairs per wavefront, ONE wavefront per SIMD): shader cycles per v_mfma_f64_16x16x4_f64.
//   MODE 0  operand reads of step k+1 in a block in front of the 16 MFMAs of step k (what the kernel does)
//   MODE 1  two reads behind each MFMA (sched_group_barrier)
//   MODE 2  no LDS reads at all (operands fixed): the issue rate of one wavefront
//   MODE 3  MODE 0 with 8 pairs per wavefront
// hipcc -O3 --offload-arch=gfx950 -o mfma_lds16 mfma_lds16.hip && ./mfma_lds16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int RS = 26, NT = 19;
template <int MODE, int PPW>
__global__ __launch_bounds__(256, 1) void k(double* out, long long* cyc, int stages) {
  extern __shared__ double s_y[];
  for (int i = threadIdx.x; i < NT * 16 * RS; i += 256) s_y[i] = 1e-3 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int rowa[PPW], rowb[PPW];
  d4 acc[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    int ti = (wave + 3 * k) % NT, tj = (wave + 5 * k + 1) % NT;
    rowa[k] = (16 * ti + (lane & 15)) * RS + (lane >> 4);
    rowb[k] = (16 * tj + (lane & 15)) * RS + (lane >> 4);
    acc[k] = d4{0, 0, 0, 0};
  }
  const int nks = 6;
  long long t0 = clock64();
  for (int st = 0; st < stages; ++st) {
    double a0[PPW], b0[PPW], a1[PPW], b1[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) { a0[k] = s_y[rowa[k]]; b0[k] = s_y[rowb[k]]; }
    for (int ks = 0; ks + 1 < nks; ks += 2) {
      const int nx = ks + 2 < nks ? 4 * ks + 8 : 4 * ks + 4;
      if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int k = 0; k < PPW; ++k) { a1[k] = s_y[rowa[k] + 4 * ks + 4]; b1[k] = s_y[rowb[k] + 4 * ks + 4]; }
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < PPW; ++k) { a0[k] = s_y[rowa[k] + nx]; b0[k] = s_y[rowb[k] + nx]; }
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k], b1[k], acc[k], 0, 0, 0);
      } else if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
          a1[k] = s_y[rowa[k] + 4 * ks + 4];
          b1[k] = s_y[rowb[k] + 4 * ks + 4];
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k], b1[k], acc[k], 0, 0, 0);
          a0[k] = s_y[rowa[k] + nx];
          b0[k] = s_y[rowb[k] + nx];
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      } else {
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < PPW; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], b0[k], acc[k], 0, 0, 0);
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int k = 0; k < PPW; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int PPW>
void run(double* out, long long* cyc, const char* name) {
  const int stages = 200, blocks = 256;
  size_t lds = (size_t)NT * 16 * RS * 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, PPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, PPW><<<blocks, 256, lds>>>(out, cyc, stages);
  hipEventRecord(e0);
  k<MODE, PPW><<<blocks, 256, lds>>>(out, cyc, stages);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)stages * 6 * PPW;
  printf("%-58s %.1f cycles per MFMA (wave 0), kernel %.3f ms -> %.2f TFLOP/s\n", name, c / nm, ms, blocks * 4 * nm * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  double* out = nullptr; long long* cyc = nullptr;
  hipMalloc(&out, (size_t)256 * 256 * 8);
  hipMalloc(&cyc, 256 * 8);
  if (!out || !cyc) { printf("allocation failed\n"); return 1; }
  run<0, 16>(out, cyc, "16 pairs, reads in a block in front of the MFMAs");
  run<1, 16>(out, cyc, "16 pairs, two reads behind each MFMA");
  run<2, 16>(out, cyc, "16 pairs, no LDS reads");
  run<3, 8>(out, cyc, "8 pairs, reads in a block");
  run<1, 8>(out, cyc, "8 pairs, two reads behind each MFMA");
  return 0;
}
