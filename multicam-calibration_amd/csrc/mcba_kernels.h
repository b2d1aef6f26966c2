// mcba_kernels.h -- launch wrappers of mcba_kernels.hip and the device buffer record sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define MCBA_REC 100  // per (frame, camera) record: W 72 | V 21 | g_f 6 | pad
#define MCBA_GP 92    // k_gram per-wavefront sums, stored [camera][k][frame block]: k = U 78 | g_c 12 | cost | pairs with data
#define MCBA_FB 40    // per frame: L 21 (diagonal slots hold 1 / L_ii) | z 6 | g_f 6 | D_f 6 | pad

#include "mcba_lm_state.h"

namespace mcba {
// Double-buffered operands (parameter slots, linearisation records) and the damping are chosen either from host
// values (lms == nullptr: idx / lam as given) or from the device-resident LM state (idx is XORed with state[3],
// lam = state[1]) -- so a whole LM iteration can be enqueued without knowing whether its trial step is accepted.
// Camera step handed to k_backsub through the kernel-argument segment (no H2D copy): 12 x C doubles, C <= 40.
struct CamStep {
  double v[480];
};

struct Sel {
  const double* lms;
  int idx;
  double lam;   // host value of the damping (lms == nullptr); lambda_min when spec != 0
  int spec;     // speculative Schur reduction of the trial linearisation (see sel_spec in mcba_kernels.hip)
  double dec;   // spec != 0: floor of Nielsen's factor the prediction assumes (mcba_lm.h: lm_spec_lambda)
  double cfl = 1.0;  // k_gram only: curvature floor of this linearisation (mcba_math.h: lm_weight) -- 1 = the IRLS weight rho', 0.1 = Triggs with a floor
};
// k_syrk's fused decision prologue (single-GPU ticks of the device-resident loop; see k_syrk in mcba_kernels.hip)
struct SyrkFuse {
  int decide;          // != 0: every workgroup sums the trial scalars and takes the accept / reject decision itself
  const double* cp0;   // per-wavefront cost sums of the two linearisation buffers (gpart + 90 nfb) ...
  const double* cp1;
  int cinner;          // ... element idx of `ncp` lives at (idx / cinner) * couter + (idx % cinner) * cstride
  int cstride;         // 4: k_gram leaves the cost of a whole workgroup (four frame blocks) in its first wavefront's slot, zeros in the others
  int cdense;          // ... up to idx % cinner == cdense; from there on every slot holds a value (k_gram_psplit ran on those frame blocks): slot = cdense cstride + (idx % cinner - cdense)
  size_t couter;
  int ncp;
  const double* bpart; // k_backsub's per-block sums (3 per block)
  int nbp;
  double* trial_out;   // 4 trial scalars for the host (cost, pred_f, |d_f|^2, |x_f|^2)
  double* lms_post;    // the state AFTER the decision (workgroup 0 writes it; later kernels of the tick read it)
  DecideArgs da;       // lam_min, lam_max, ftol, xtol
  const double* timeout_word;  // device word a back-substitution workgroup of k_solve_backsub stamps with its tick's number when its poll ran out
  double seq_prev;             // the previous tick's number: found in the word, this tick's trial point is stale -> rebuild only
};
// k_solve_cam (mcba_solve.hip): reduced camera system factorised and solved by one workgroup
struct SolveArgs {
  const double* red;         // S0 (n x n) | rhs | diag U | g_c | 16 scalars
  const double* lms_in;      // device LM state as the tick's decision left it (== lms unless k_syrk took the decision)
  double* lms;               // device LM state the NEXT tick starts from (written back here)
  double* work;              // npad x npad scratch (used when the factor does not fit LDS)
  double* dc;                // n doubles: the camera step
  const double* x0;          // parameter slots (camera block first)
  const double* x1;
  const unsigned char* fixed;  // n flags (1 = parameter held fixed) or nullptr
  const double* dscale;      // numeric x_scale: D_c = dscale[0 .. n) instead of diag(U), or nullptr
  double* host_state;        // host-mapped ring slot of MCBA_LMS doubles, or nullptr
  double* flag;              // k_solve_backsub: device word released with `seq` when the camera step is in place (else nullptr)
  const double* timeout_word;  // see SyrkFuse (one-collective ticks: the decision taken here checks it); may be nullptr
  double seq;
  double gtol, lam_max;
  int n, npad, use_lds;
  int cw;                    // camera block width: 12, or 6 = the intrinsics of every camera are held fixed (row i of the system is parameter 6 + i % 6 of camera i / 6)
  // decide != 0 (frame-sharded ticks with one collective): the all-reduced trial scalars sit behind the system and the
  // accept/reject decision is taken HERE, then checked against the prediction the speculative Schur reduction was built on
  int decide;
  double lam_min, ftol, xtol;
  double dec_floor;          // floor of Nielsen's damping factor (0 = 1/3): used by the decision taken here and to check the prediction
  double stage_tag;          // > 9 cameras: this launch's number in the handle's life (never repeats: the stager workgroups release it, workgroup 0 waits for it)
};
void launch_transpose_obs(hipStream_t st, const double* raw, double* obs_t, int C, int F, int N, int Fpad);
void launch_gram(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, Sel s, const double* x0, const double* x1, double* rec0, double* rec1, double* gp0, double* gp1, int C, int N, int Fpad, int split,
                 int planar = 0,   // planar: every board point has z = 0 exactly (with f_scale = 1 the fused kernel's FAST instance runs)
                 double* chunk = nullptr, int nchunk = 0,   // split == 3: scratch for the point-chunk tail (gram_chunk_doubles) and the number of chunks
                 int npw = 4,                               // split == 4 / 5 (point split inside the workgroup): wavefronts per (camera, frame block), 4 or 2
                 int cw = 12,                               // camera block width: 12, or 6 = intrinsics held fixed (role A alone: split 4 = point split, anything else = the role-A half of the split roles)
                 int slots = 1024,                          // wavefront slots of the device (4 x CUs): where the launch variants cut a shard into rounds
                 const double* ltab = nullptr);             // loss == LOSS_TABLE: three planes [C][N][Fpad] of (u, v) pairs -- 0.5 f_scale^2 rho, rho', J_scale^2 (mcba_set_loss_table)
void gram_time_next_launch(hipEvent_t start, hipEvent_t stop);  // measurement: the next fused k_gram launch of this thread carries the events on its dispatch (the kernel's own begin / end)
bool gram_time_pending();                                       // ... still pending after the launch: another variant ran, bracket it the usual way
int gram_round_blocks(int C, int nfb, int slots);   // frame blocks (a multiple of 4) that whole rounds of the wavefront slots cover; split 2 / 3 / 5 handle the rest as a tail
size_t gram_psplit_lds_bytes(int npw, int cw = 12);  // dynamic LDS of k_gram_psplit
int gram_psplit_set_lds_limit();             // raises the dynamic-LDS limit of its instances (0 = ok)
size_t gram_chunk_doubles(int C, int nfb, int nchunk, int slots);   // doubles of that scratch for C cameras x nfb frame blocks
void launch_cost(hipStream_t st, int loss, double f_scale, const double* obs_t, const double* obj, const double* x, double* cpart, double* res, int C, int F, int N, int Fpad, int nch,
                 double fill = 0.0);   // what the residual vector holds where a scalar is missing: 0, or NaN (then the vector carries its own row mask)
size_t syrk_lds_bytes(int C, int FS, int cw = 12);
void launch_syrk(hipStream_t st, Sel s, const SyrkFuse& fz, const double* rec0, const double* rec1, double* fbuf, double* fpart, const int* tile_i, const int* tile_j, double* spart, int C, int F, int Fpad, int NT, int NP, int G, int sq, int sr, int FS, int ppw,
                 const double* dscale = nullptr,   // dscale: D = 1 / x_scale^2 in the layout of x (numeric x_scale), nullptr: D = diag(J^T J)
                 int cw = 12);                     // camera block width (6: intrinsics held fixed; the 4-tile variant only)
int syrk_items_per_thread();
// bpart != nullptr (speculative frame-sharded ticks): the trial scalars are summed here as well (red + nsys .. + 8) and the LM
// state is copied to state_copy (MCBA_LMS doubles)
void launch_reduce_system(hipStream_t st, Sel s, const double* gp0, const double* gp1, const double* spart, const double* fpart, const int* tile_i, const int* tile_j, double* red, int C, int nfb, int G, int NT, int NP, int nfblocks, int rank_slot,
                          const double* bpart = nullptr, int nbp = 0, double* state_copy = nullptr, int cw = 12,
                          const double* timeout_word = nullptr, double seq_prev = 0.0);  // (speculative ticks: trial scalar 5 = the previous fused back-substitution of this shard gave up)
void launch_backsub(hipStream_t st, Sel s, const double* rec0, const double* rec1, const double* fbuf, const CamStep& dc, double* x0, double* x1, double* bpart, int C, int F, int Fpad, int cw = 12);
void launch_backsub_dev(hipStream_t st, Sel s, const double* rec0, const double* rec1, const double* fbuf, const double* dc_dev, double* x0, double* x1, double* bpart, int C, int F, int Fpad, int cw = 12);
size_t solve_lds_bytes(int npad, int use_lds);
int solve_fits_lds(int npad, int lds_limit);
int solve_set_lds_limit(int npad, int use_lds);
void launch_solve_cam(hipStream_t st, const SolveArgs& a);
// solve + the back-substitution of the next trial step in one launch (a.use_lds variants, a.flag set); early_state = the LM state
// the tick's decision left (final as far as the slot bit goes), or -- spec != 0, the solve decides -- a copy of the state before it
int solve_backsub_set_lds_limit(int npad, int cw = 12);
void launch_solve_backsub(hipStream_t st, const SolveArgs& a, Sel sl, const double* rec0, const double* rec1, const double* fbuf, double* x0, double* x1, double* bpart, int C, int F, int Fpad,
                          const double* early_state, int max_polls, int spec, double* timeout_dev, double* timeout_host,   // (a.cw selects the camera block width)
                          int strict = 0);  // != 0: the waiting workgroups acquire the release word with an agent-scope fence (mcba_backsub.h: release_word_acquired)
void launch_sum_trial(hipStream_t st, Sel s, const double* cp0, const double* cp1, int cstride, int cinner, size_t couter, int ncp, const double* bpart, int nbp, double* out, DecideArgs da);
void launch_decide(hipStream_t st, const double* trial8, DecideArgs da);
void launch_lm_init(hipStream_t st, const double* red_scal, double* lms, double lam0, int sel, double cfl, double cfl_switch, double* clear8 = nullptr);  // mcba_lm_run: the start state, on the device
void launch_pack_result(hipStream_t st, const double* x, const double* gc, const double* fbuf, const unsigned char* fixed, double* out, int C, int F, int cw);  // mcba_lm_result
void launch_jacobian(hipStream_t st, int loss, double f_scale, const double* obs_raw, const double* obj, const double* x, double* jac, double* res, int C, int F, int N, int Fpad, int robust);
int syrk_set_lds_limit(size_t bytes);
// pre-filter, frame subsets, undistortion, reprojection diagnostics (mcba_diag.hip)
void launch_frame_err(hipStream_t st, const double* obs_t, const double* obj, const double* x, double* err, double* mean_cf, double* full_cf, int C, int F, int N, int Fpad,
                      void* prefilter_state = nullptr);   // non-NULL: the launch also zeroes the selection's state (launch_prefilter_select(..., state_cleared = true) follows)
size_t select_state_bytes(int groups);  // per group: u64 prefix, rank, count, value (bit pattern of the selected double) + a 256-bin histogram
void launch_select(hipStream_t st, const double* v, const unsigned char* fmask, size_t per_group, int groups, int Fpad, void* sel, int upper,
                   int skey = 0);   // != 0: values of either sign (compared through an order-preserving key); 0: values >= +0 (the pre-filter's errors)
int launch_select_hist(hipStream_t st, const double* v, const unsigned char* fmask, size_t per_group, int Fpad, void* sel, unsigned long long prefix, int pass, unsigned int* hist256);
// the pre-filter's selection on the device (mcba_prefilter): see mcba_diag.hip
size_t prefilter_state_bytes();
void launch_prefilter_select(hipStream_t st, const double* err, const double* mean_cf, const double* full_cf, unsigned char* fmask, unsigned char* status, double* worst, void* state,
                             unsigned char* packed, int C, int F, int N, int Fpad, double threshold, bool state_cleared = false);
void launch_prefilter_status(hipStream_t st, unsigned char* status, const double* worst, void* state, unsigned char* packed, int F, double threshold, int from_state);
void launch_gather_params(hipStream_t st, const double* x_src, const int* frames, double* x_dst, int C, int Fdst);
void launch_clip(hipStream_t st, double* x, const double* lo, const double* hi, size_t n);  // x <- min(max(x, lo), hi): the trial point of a bounded step
bool launch_store_small(hipStream_t st, double* dst, const double* src_host, size_t n);  // <= 480 doubles through the kernel-argument segment (no blocking copy); false: too many
double measure_fp64_issue_rate(int ncu);  // TFLOP/s of independent v_fma_f64 at one wavefront per SIMD on `ncu` compute units (measurement aid)
void launch_gather_frames(hipStream_t st, const double* src_raw, const int* frames, double* dst_raw, int C, int Fsrc, int Fdst, int N,
                          const int* only_cam = nullptr);   // non-NULL: destination frame j keeps the detection of camera only_cam[j] alone (NaN for the others)
void launch_seen_bits(hipStream_t st, const double* obs_raw, size_t count, unsigned long long* words);  // words: ceil(count / 64) of them
void launch_undistort(hipStream_t st, const double* uv, double* out, size_t n, const double* K4, const double* dist5, int iters);
void launch_reproj_diag(hipStream_t st, const double* obs_t, const double* obj, const double* x, const double* dist5, const double* bn, double* und, double* repro, double* trans, double* err, int C, int F, int N,
                        int Fpad, int iters, int lm_iters);
// calibrate()'s per-view work and pose graph (mcba_pnp.hip).  mode 0: homographies of the listed views (out: nviews x 9); mode 1: board poses --
// views != nullptr: of the listed (camera, frame) views (out: nviews x 6), else of every (camera, frame) (out (C,F,6) and / or poses_t [C][6][Fpad])
void launch_view_complete(hipStream_t st, const double* obs_t, unsigned char* out, int C, int F, int N, int Fpad);
void launch_pnp(hipStream_t st, int mode, const double* obs_t, const double* obj, const double* intr9, const int* views, int nviews, const double* bn3, int C, int F, int N, int Fpad, int und_iters, int lm_iters,
                double* out, double* poses_t, unsigned char* valid, unsigned char* nit);
// poses addressed as p[c * sc + f * sf + k * sk]; rel [n_edges][6][Fpad]; world [C][6][Fpad] scratch; out (F, 6)
void launch_zhang(hipStream_t st, const double* H, const unsigned char* ok, const int* views, int nviews, const double* sizes, int C, double* intr9, unsigned char* closed);
void launch_pose_pairs(hipStream_t st, const double* poses, size_t sc, size_t sf, size_t sk, const int* edges, int n_edges, int F, int Fpad, double* rel);
void launch_pose_chain(hipStream_t st, const void* sel, size_t sel_state_bytes, const int* edges, int n_edges, int root, int C, double* ext, double* transforms, double* counts);
void launch_pose_consensus(hipStream_t st, const double* poses, size_t sc, size_t sf, size_t sk, const double* ext, int C, int F, int Fpad, double* world, double* out);
// triangulation (mcba_triangulate.hip): up to 8 cameras; P = K [R | t] row-major 3x4, K = (fx, fy, cx, cy), dist = (k1 k2 p1 p2 k3)
struct TriCams {
  double P[8][12];
  double K[8][4];
  double dist[8][5];
};
// single-camera calibration with the 5-coefficient model (mcba_calib.hip): per view the 15 x 15 Gauss-Newton block (upper triangle, 120), the gradient (15), the cost
void launch_calib_views(hipStream_t st, const double* uvs, const double* obj, const double* intr9, const double* poses, int V, int N, double* out);
int launch_triangulate(hipStream_t st, int C, const double* uvs, const TriCams& cams, double* out, size_t npts, int iters);
// 9 .. 64 cameras: cams_dev = C x {P[12], K[4], dist[5]} doubles in device memory; one wavefront per point
int launch_triangulate_wave(hipStream_t st, int C, const double* uvs, const void* cams_dev, double* out, size_t npts, int iters);
}  // namespace mcba
