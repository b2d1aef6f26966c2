// mcba_kernels.h -- launch wrappers of mcba_kernels.hip and the device buffer record sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define MCBA_REC 100  // per (frame, camera) record: W 72 | V 21 | g_f 6 | pad
#define MCBA_GP 92    // per gram wavefront partial: U 78 | g_c 12 | cost | n_pairs_with_data
#define MCBA_FB 40    // per frame: L 21 | z 6 | g_f 6 | D_f 6 | pad

namespace mcba {
void launch_transpose_obs(hipStream_t st, const double* raw, double* obs_t, int C, int F, int N, int Fpad);
void launch_gram(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, const double* x, double* rec, double* gpart, int C, int N, int Fpad, int split);
void launch_cost(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, const double* x, double* cpart, double* res, int C, int F, int N, int Fpad, int nch);
size_t syrk_lds_bytes(int C, int FS);
void launch_frame_factor(hipStream_t st, const double* rec, double* fbuf, double* fpart, int C, int F, int Fpad, double lambda);
void launch_syrk(hipStream_t st, const double* rec, const double* fbuf, const int* tile_i, const int* tile_j, double* spart, int C, int F, int Fpad, int NT, int NP, int G, int fpc, int FS, int ppw);
int syrk_items_per_thread();
void launch_reduce_system(hipStream_t st, const double* gpart, const double* spart, const double* fpart, double* red, int C, int nfb, int G, int NT, int NP, int nfblocks, int rank_slot);
void launch_backsub(hipStream_t st, const double* rec, const double* fbuf, const double* dc, const double* xs, double* xd, double* bpart, int C, int F, int Fpad, double lambda);
void launch_sum_trial(hipStream_t st, const double* cpart, int cstride, int ncp, const double* bpart, int nbp, double* out);
void launch_jacobian(hipStream_t st, int loss, double f_scale, const double* obs_raw, const double* obj, const double* x, double* jac, double* res, int C, int F, int N, int Fpad, int robust);
int syrk_set_lds_limit(size_t bytes);
}  // namespace mcba
