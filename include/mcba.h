/* mcba.h -- C ABI of libmcba.so, the MI355X (gfx950) bundle-adjustment hot path.
 *
 * The reference (dattalab-6-cam/multicam-calibration) is pure Python and has no FFI: the path sits
 * behind two Python seams.  This header is the boundary a maintainer would
 * bind with ctypes in place of them (INTEGRATION.md shows the stub):
 *
 *   outer seam  multicam_calibration/bundle_adjustment.py:195-204,307-327  bundle_adjust(...)
 *   inner seam  scipy least_squares fun/jac callables:
 *               residuals()                       bundle_adjustment.py:66-98
 *               bundle_adjustment_sparsity()      bundle_adjustment.py:101-125  (structure is implicit here)
 *               scipy 2-point FD Jacobian         scipy/optimize/_numdiff.py:628-705 (replaced by analytic blocks)
 *               scipy TRF + LSMR linear algebra   scipy/optimize/_lsq/trf.py:401-560 (replaced by LM + Schur, incl. the
 *                                                 reduced solve and the termination tests: mcba_lm_auto_*)
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 = MCBA_OK), never throws;
 *     mcba_last_error() returns the message of the last failure on the calling thread.
 *   - parameter vector x: [C x (fx fy cx cy k1 k2 rx ry rz tx ty tz) | F x (rx ry rz tx ty tz)]  float64,
 *     exactly serialize_params() of the reference (bundle_adjustment.py:128-157).
 *   - observations: float64 (C,F,N,2) C-order, NaN = missing scalar (bundle_adjustment.py:82-84,97).
 *   - one handle = one GPU = one process (frames are sharded across processes by the caller);
 *     calls on a handle come from one host thread.  The library owns its device buffers until
 *     mcba_destroy(); host arrays belong to the caller.
 *   - all kernels are enqueued on the handle's stream (mcba_set_stream; default: the null stream) and
 *     functions that return host data synchronise that stream.
 */
#ifndef MCBA_H
#define MCBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCBA_OK 0
#define MCBA_ERR_HIP 1        /* a HIP runtime call failed */
#define MCBA_ERR_ARG 2        /* bad argument / wrong call order */
#define MCBA_ERR_NONFINITE 3  /* a used observation projects to NaN/inf (scipy: "Residuals are not finite") */
#define MCBA_ERR_NODEVICE 4   /* no usable gfx950 device */

/* scipy.optimize.least_squares `loss=` (least_squares.py:160-227) */
#define MCBA_LOSS_LINEAR 0
#define MCBA_LOSS_SOFT_L1 1
#define MCBA_LOSS_HUBER 2
#define MCBA_LOSS_CAUCHY 3
#define MCBA_LOSS_ARCTAN 4
#define MCBA_LOSS_TABLE 5   /* not a name for mcba_set_loss: what mcba_set_loss_table switches the handle to (least_squares' callable `loss`) */

typedef struct mcba_handle mcba_handle;
typedef struct mcba_buffer mcba_buffer;   /* a device array that outlives its handle (mcba_residuals_detach) */

/* ---- library ------------------------------------------------------------------------------- */
int mcba_abi_version(void);            /* 7.  Bumped when this header changes: 7 (round 6) ADDS the mcba_calib_* (incl. mcba_calib_start, mcba_calib_graph) / mcba_pose_* / mcba_create_views block below (calibrate() on the device) and leaves every ABI-6 entry point as it was; 6 (round 5) ADDED mcba_prefilter, mcba_prefilter_subset, mcba_lm_run, mcba_lm_history, mcba_lm_result, mcba_set_bounds / _frozen, mcba_set_loss_table, mcba_set_trial, mcba_calib_normal_equations (and the diagnostics / life-cycle helpers declared below as ABI 6) and leaves every
                                        * ABI-5 entry point as it was.  (ABI 5 gave LM-state slots 25 / 26 -- "reserved" before -- their meaning: curvature floor / switch
                                        * fraction; a caller that zeroes them gets the handle's floor, fixed.) */
const char* mcba_last_error(void);
int mcba_device_count(int* count);

/* ---- problem life cycle -------------------------------------------------------------------- */
/* n_cameras C, n_frames F (local shard), n_points N, HIP device ordinal. */
int mcba_create(mcba_handle** out, int n_cameras, int n_frames, int n_points, int device);
int mcba_destroy(mcba_handle* h);
/* Device and pinned buffers of destroyed handles are parked in a per-process pool and handed out again (hipMalloc / hipFree are
 * synchronising driver calls: 4 ms of a 16 ms bundle_adjust() at 6 x 10 000 x 54 before the pool).  MCBA_POOL_MB caps what is
 * parked (default 2048; 0 = no pool); this returns everything to the driver.  The solver's own buffers (records, partial sums,
 * reduce buffer, state ring) are allocated on the first call that needs them: a handle that only runs the pre-filter over all
 * frames of a long recording holds the observations and nothing else. */
int mcba_pool_trim(void);
/* Bytes of device memory the handle holds right now. */
size_t mcba_device_bytes(const mcba_handle* h);
/* Returns every device buffer to the pool except what defines the problem (observations in both layouts, board, parameter slots, the box of
 * mcba_set_bounds): solver buffers, pre-filter scores, Jacobian / residual blocks, calibrate()'s poses.  Calls that need them allocate them
 * again; a numeric x_scale / frozen set given before the trim is put back into the new buffers.  For handles that are parked so that
 * something can be produced from them later (the lazily attached result.jac): 0.26 -> 0.11 GB at 6 x 10 000 x 54.  Synchronises. */
int mcba_trim(mcba_handle* h);
/* The observations the handle holds, (C,F,N,2) doubles, back to the host (the values the solve saw). */
int mcba_download_observations(mcba_handle* h, double* uvs);
/* hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream); NULL = null stream. */
int mcba_set_stream(mcba_handle* h, void* hip_stream);
/* Observations (C,F,N,2) and board points (N,3), host pointers.  Re-laid out on the GPU as
 * [camera][point][frame] (u,v) pairs so that a wavefront reads 64 consecutive frames coalesced. */
int mcba_upload_observations(mcba_handle* h, const double* uvs, const double* objpoints);
/* Robust loss and f_scale of least_squares (default soft_l1, 1.0: bundle_adjustment.py:301-303). */
int mcba_set_loss(mcba_handle* h, int loss, double f_scale);
/* least_squares' CALLABLE `loss` -- rho(z) -> (rho, rho', rho'') (least_squares.py:160-227; the reference forwards it untouched: bundle_adjustment.py:301-313).
 * The function is the caller's, so are its values: tab3 = three (C,F,N,2) arrays in the order of the observations, evaluated at the residuals
 * (mcba_residuals) of the point that is linearised next: [0] 0.5 f_scale^2 rho(z), [1] rho'(z), [2] max(rho' + 2 rho'' z, EPS) (scipy's J_scale^2,
 * common.py:720-731); entries of missing observations are ignored.  mcba_linearize then builds the normal equations with these weights; the
 * trial cost of mcba_step is the caller's to evaluate (its function on the trial residuals); mcba_jacobian_eval returns unscaled rows;
 * mcba_step_linearize and the device-resident loops (mcba_lm_iterate, mcba_lm_auto_*, mcba_lm_run) refuse to run.  mcba_set_loss switches back. */
int mcba_set_loss_table(mcba_handle* h, const double* tab3);

/* Camera block width (round 4, ABI 5).  12 (default): all 12 parameters of every camera are variables, as in the reference
 * (bundle_adjustment.py:113,121-122,149-155).  6: the intrinsics (fx fy cx cy k1 k2) of EVERY camera are held fixed -- BASELINE
 * configs[1] ("intrinsics fixed: extrinsics + points only"; SURVEY section 8c-8: the reference has no entry point for it, its oracle is
 * a wrapper fun(y) = residuals(scatter(y, frozen intrinsics))).  With 6 the linearisation accumulates the (rho, t) blocks alone,
 * the reduced camera system is 6C x 6C -- row i is parameter 6 + i % 6 of camera i / 6 -- and EVERY camera-system quantity of this
 * ABI has 6 entries per camera: mcba_reduced_size / mcba_get_reduced (n = 6C), the camera step of mcba_step* / mcba_lm_* /
 * mcba_get_cam_step, the `fixed` flags of mcba_lm_auto_config.  The parameter vector x keeps the reference's layout (12 per camera;
 * the intrinsics are copied unchanged into every trial point).  Must be called before the first solver entry point of the handle
 * (MCBA_ERR_ARG afterwards, and for width 6 with more than 26 cameras: hold those with the `fixed` flags instead). */
int mcba_set_camera_block(mcba_handle* h, int width);
int mcba_get_camera_block(const mcba_handle* h);
/* Curvature weight of the linearisations enqueued from now on: w = max(rho' + 2 rho'' f^2, floor * rho'), 0 < floor <= 1.
 * 1 (default) = the IRLS weight rho' (monotone: the model for points far from the optimum); 0.1 = Triggs' second-order term with a
 * safety floor (fast at the optimum).  The reference has no counterpart: scipy's TRF always uses the Triggs weight clamped at EPS
 * (common.py:720-731); the stationary point does not depend on the choice, the number of evaluations does (16 -> 6 to the reference's
 * default tolerance at 6 x 10 000 x 54).  solver.py switches between ticks: IRLS, Triggs after an accepted step that gained < 1 %. */
int mcba_set_curvature_floor(mcba_handle* h, double floor);
double mcba_get_curvature_floor(const mcba_handle* h);

/* least_squares' numeric `x_scale` (forwarded verbatim by the reference: bundle_adjustment.py:301-313; scipy least_squares.py
 * :243, trf.py:415-420): 12C + 6F positive doubles in the layout of x -> the FIXED damping matrix D = diag(1 / x_scale^2) replaces
 * Marquardt's D = diag(J^T J) (which is x_scale = 'jac', the reference's default) in the frame blocks, the camera block and the
 * predicted reduction.  NULL returns to 'jac'.  MCBA_ERR_ARG with scipy's message if an entry is not positive and finite. */
int mcba_set_x_scale(mcba_handle* h, const double* x_scale);
/* Box constraints and the working set of an active-set method (round 5; the reference forwards `bounds` to scipy's least_squares:
 * bundle_adjustment.py:301-313, scipy trf_bounds).
 * mcba_set_bounds: lo <= x <= hi, 12C + 6F doubles each in the layout of x (-inf / +inf = none; both NULL = no constraints).  From then
 *   on the trial point of every mcba_step / mcba_step_linearize / mcba_step_fetch is projected onto the box before its cost is taken;
 *   the device-resident loops (mcba_lm_iterate, mcba_lm_auto_*, mcba_lm_run) refuse to run while bounds are set.
 * mcba_set_frozen: mask of 12C + 6F bytes (non-zero = frozen; NULL = none) -- a frozen FRAME coordinate gets a step of exactly 0 and
 *   nothing couples to it in the Schur reduction and the back-substitution (the reported frame gradient stays the true one); camera
 *   entries are ignored here (freeze camera parameters with the flags of mcba_lm_auto_config, or leave their rows out of a host solve).
 *   Takes effect with the next mcba_build_reduced. */
int mcba_set_bounds(mcba_handle* h, const double* lo, const double* hi);
int mcba_set_frozen(mcba_handle* h, const unsigned char* mask);

/* ---- parameter slots (two flat vectors live on the GPU: 0 and 1) ----------------------------- */
int mcba_set_params(mcba_handle* h, int slot, const double* x);   /* 12C+6F doubles, host */
int mcba_get_params(mcba_handle* h, int slot, double* x);
int mcba_copy_params(mcba_handle* h, int dst_slot, int src_slot);  /* device to device */

/* ---- residual / Jacobian evaluation (replaces residuals() and scipy's FD Jacobian) ----------- */
/* Robust cost 0.5*f_scale^2*sum rho((f/f_scale)^2) at x[slot]; *n_residuals = number of non-NaN scalars.
 * MCBA_ERR_NONFINITE if the cost is not finite. */
int mcba_cost(mcba_handle* h, int slot, double* cost, double* n_residuals);
/* Dense residual array (C,F,N,2), observed - predicted, 0 where the observation is NaN (host). */
int mcba_residuals(mcba_handle* h, int slot, double* res);
/* The same array -- with NaN instead of 0 where the observation is missing, so that it carries its own row mask (ABI 6) -- left
 * ON THE DEVICE as an object of its own (it outlives the handle): api.bundle_adjust hands it to the
 * OptimizeResult and downloads it when `result.fun` is first read -- scipy materialises `fun` (trf.py:557-560), but 52 MB of
 * device-to-host copy at 6 x 10 000 x 54 for a field few callers read was the largest single item of the call.
 * mcba_buffer_count: doubles in it; mcba_buffer_download: copy to host (synchronises); mcba_buffer_free: back to the pool. */
int mcba_residuals_detach(mcba_handle* h, int slot, mcba_buffer** out);
size_t mcba_buffer_count(const mcba_buffer* b);
int mcba_buffer_download(mcba_buffer* b, double* host);
int mcba_buffer_free(mcba_buffer* b);
/* Which scalars of the uploaded observations are present (not NaN), from the GPU's own copy: bits = numpy.packbits(~numpy.isnan(uvs))
 * over the (C, F, N, 2) array, (2 C F N + 7) / 8 bytes, host.  It is the row selection of the reference's residual vector and
 * Jacobian (bundle_adjustment.py:68-69 `mask = ~np.isnan(uvs)`, :101) without a pass over the caller's array; synchronises. */
int mcba_seen_bits(mcba_handle* h, unsigned char* bits);
/* Materialise residuals + analytic Jacobian blocks on the GPU: per scalar residual 18 doubles
 * [12 camera columns | 6 frame-pose columns] in (C,F,N,2,18) order = the CSR `data` array of the
 * reference's Jacobian when no observation is missing.  robust_scaled != 0 applies scipy's
 * sqrt(rho' + 2 rho'' f^2) row scaling (common.py:720-731).  Result stays on the GPU. */
int mcba_jacobian_eval(mcba_handle* h, int slot, int robust_scaled);
/* Copy the last mcba_jacobian_eval() result to the host: jac (C,F,N,2,18), res (C,F,N,2); either may be NULL. */
int mcba_jacobian_download(mcba_handle* h, double* jac, double* res);

/* ---- Levenberg-Marquardt building blocks (replace trf.py:450-551 + LSMR) ---------------------- */
/* Linearise at x[slot]: per (camera, frame) local Gram matrices expanded to W_cf (12x6), V_cf (6x6),
 * g_f and per-wave partial sums of U_c (12x12), g_c and the robust cost.  Kept on the GPU. */
int mcba_linearize(mcba_handle* h, int slot);
/* Schur-reduce the current linearisation with damping lambda (frame blocks damped by
 * lambda*diag(V_f)) into the reduce buffer (device), laid out as doubles, n = 12C:
 *   [0,n*n)      S0  = blockdiag(U_c) - sum_f W_f (V_f + lambda D_f)^-1 W_f^T      (undamped camera block)
 *   [n*n,+n)     rhs = -g_c + sum_f W_f (V_f + lambda D_f)^-1 g_f
 *   [..,+n)      diag(U)          [..,+n)  g_c
 *   [..,+16)     scalars: 0 cost, 1 number of (camera, frame) pairs with data, 2 n_cholesky_failures, 3 reserved,
 *                         4..15 per-rank slots: max |g_f| of THIS shard goes to slot 4+rank_slot, others 0
 * so that one all-reduce(SUM) of the whole buffer over the frame shards gives the global system. */
int mcba_build_reduced(mcba_handle* h, double lambda, int rank_slot);
size_t mcba_reduced_size(const mcba_handle* h);          /* doubles in the reduce buffer: system + 8 trial scalars + MCBA_LM_STATE */
/* Let the caller own the reduce buffer (device pointer, mcba_reduced_size() doubles), e.g. a torch
 * tensor handed to torch.distributed.all_reduce (RCCL).  NULL restores the internal buffer. */
int mcba_bind_reduce_buffer(mcba_handle* h, double* device_ptr);
int mcba_get_reduced(mcba_handle* h, double* host);      /* D2H of the system part (n*n+3n+16 doubles) */
/* Back-substitute: given the camera step (12C doubles, host) solve every frame step,
 * write x[dst_slot] = x[src_slot] + delta and evaluate the robust cost there.  Trial scalars
 * (8 doubles, appended to the reduce buffer at offset n*n+3n+16):
 *   0 cost(x_dst)  1 sum_f d_f^T (lambda D_f d_f - g_f)  2 sum |d_f|^2  3 sum |x_f|^2  4 n_residuals  5..7 reserved */
int mcba_step(mcba_handle* h, const double* delta_cam, double lambda, int src_slot, int dst_slot);
/* Same, but instead of a cost-only pass the trial point x[dst] is LINEARISED (as mcba_linearize) into the
 * handle's second linearisation buffer; its robust cost fills trial scalar 0 (scalar 4 = number of
 * (camera, frame) pairs with data).  If the step is accepted, mcba_accept_linearization() makes that buffer the
 * current one at no GPU cost -- an accepted LM iteration then needs ONE pass over the observations, not two. */
int mcba_step_linearize(mcba_handle* h, const double* delta_cam, double lambda, int src_slot, int dst_slot);
int mcba_accept_linearization(mcba_handle* h);
int mcba_get_trial(mcba_handle* h, double* host8);
int mcba_set_trial(mcba_handle* h, const double* host8);   /* the eight trial scalars, written back (a tabulated loss in a frame-sharded run: [the caller's cost of THIS shard's trial point, the step's other scalars], before the all-reduce) */
/* Convenience for the single-GPU loop (one ABI crossing instead of two):
 *   mcba_reduce_fetch     = mcba_build_reduced + mcba_get_reduced
 *   mcba_step_fetch       = mcba_step (linearize == 0) or mcba_step_linearize (linearize != 0) + mcba_get_trial */
int mcba_reduce_fetch(mcba_handle* h, double lambda, int rank_slot, double* host);
int mcba_step_fetch(mcba_handle* h, const double* delta_cam, double lambda, int src_slot, int dst_slot, int linearize, double* host8);
/* ---- device-resident LM iteration: ONE host synchronisation per iteration ----------------------------
 * The reduce buffer carries, behind the system (n*n+3n+16) and the 8 trial scalars, MCBA_LM_STATE = 32 doubles:
 *   0 cost  1 lambda  2 nu  3 sel (index of the current parameter slot AND linearisation buffer)  4 accepted
 *   5 cost_new  6 predicted reduction  7 ratio  8 step norm  9 x norm  10 actual reduction;
 *   11..24 belong to the device-resident solve below (0 for the host-solve entry points);
 *   25 curvature floor of the NEXT linearisations (1 = the IRLS weight rho', 0.1 = Triggs' term with a floor; 0 in the array handed to
 *      mcba_lm_set_state = the handle's value, mcba_set_curvature_floor)   26 switch fraction (> 0: the decision moves slot 25 -- Triggs
 *      after an accepted step that gained less than this fraction of the cost, IRLS after a rejection whose cost rose by more than 1e-9 of
 *      itself; 0: slot 25 stays)   27..31 reserved.
 * mcba_lm_set_state uploads it (after mcba_linearize(slot = sel) + mcba_build_reduced);
 * mcba_lm_trial            : back-substitute delta_cam from the current point, linearise the trial point
 *                            (other slot / buffer), sum its cost  -> trial scalars;  [all-reduce them when sharded]
 * mcba_lm_decide_reduce    : accept/reject (with a round-off guard for |dF| <= 32 EPS F) + Nielsen damping update (a rejection whose
 *                            cost rose by less than 1e-9 of itself doubles the damping without escalating) ON THE
 *                            GPU (same rule as the host driver, solver.py),
 *                            then Schur-reduce whichever linearisation is now current with the new lambda;
 *                            pred_cam = d_c^T(lambda D_c d_c - g_c), dcn2 = |d_c|^2, xcn2 = |x_c|^2 from the host solve;
 *                            [all-reduce the system when sharded]
 * mcba_lm_fetch            : D2H of system + trial scalars + state (mcba_reduced_size() doubles) and synchronise;
 * mcba_lm_iterate          : the three above in one call (single GPU);
 * mcba_lm_rebuild          : Schur-reduce again with the state's lambda (after the host changed it with set_state). */
#define MCBA_LM_STATE 32
int mcba_lm_set_state(mcba_handle* h, const double* state);
int mcba_lm_trial(mcba_handle* h, const double* delta_cam);
int mcba_lm_decide_reduce(mcba_handle* h, double pred_cam, double dcn2, double xcn2, double lam_min, double lam_max, int rank_slot);
int mcba_lm_rebuild(mcba_handle* h, int rank_slot);
int mcba_lm_fetch(mcba_handle* h, double* host);
int mcba_lm_iterate(mcba_handle* h, const double* delta_cam, double pred_cam, double dcn2, double xcn2, double lam_min, double lam_max, double* host);
/* ---- device-resident LM loop: NO host synchronisation per iteration --------------------------------------------
 * The reduced camera system is factorised and solved on the GPU as well (k_solve_cam: blocked FP64 Cholesky on
 * v_mfma_f64_16x16x4, one workgroup), the ftol / xtol / gtol tests of scipy (least_squares.py / trf.py:529-551) run
 * there, and the host only enqueues "ticks" and reads the state each tick posts to a host-mapped ring of 16 slots.
 *   state 11 pred_cam  12 |d_c|^2  13 |x_c|^2  14 skip (reduced solve failed: next tick only re-damps)  15 done (0 or
 *   the scipy status 1..4)  16 first-order optimality  17 trial evaluations  18 accepted steps  19 pending verdict
 *   20 lambda of the last trial  21 cost before it  22 ticks that did work  23 solve info (0 ok, 1 not positive
 *   definite, 2 a frame block failed)  24 last tick was a rebuild  31 (ring slots) sequence number.
 * mcba_lm_auto_config : tolerances, damping bounds, optional mask (n bytes, 1 = camera parameter held fixed);
 * mcba_lm_auto_solve  : optimality + termination verdict + solve of the system in the reduce buffer -> camera step on
 *                       the device, state posted to ring slot seq % 16 with sequence number `seq` (>= 1);
 *                       call once after mcba_build_reduced + mcba_lm_set_state (decide = 0), then once per tick;
 * mcba_lm_auto_trial  : back-substitute that step, linearise the trial point, sum its cost; decide > 0: accept/reject on
 *                       the same launch (single GPU);  [sharded: decide = 0, all-reduce the 8 trial scalars];  decide = -1:
 *                       no sums here -- the speculative reduction that follows (mcba_lm_auto_reduce(h, 2, .)) writes the trial
 *                       scalars itself, one launch less per tick;
 * mcba_lm_auto_reduce : decide = 0: Schur reduction of the current linearisation (the decision has been taken);
 *                       decide = 1: the stand-alone decision kernel first (sharded runs with TWO collectives: trial scalars,
 *                       then the system);  decide = 2: SPECULATIVE reduction (sharded runs with ONE collective): the trial
 *                       linearisation is reduced before the decision is known, on the prediction "accepted, lambda' =
 *                       max(lambda / 3, lambda_min)", and the 8 trial scalars [cost, pred_f, |d_f|^2, |x_f|^2, #pairs, stale, 0, 0]
 *                       are (re)written behind the system (stale = 1 if a back-substitution workgroup of this handle's previous fused
 *                       launch gave up waiting for its solve: the SUM over the shards is what the deciding solve tests, so that every
 *                       shard discards such a tick); the caller all-reduces [system | trial scalars] in one go
 *                       (offset 0, n*n + 3n + 16 + 8 doubles) and calls mcba_lm_auto_solve(seq, decide = 1), which takes
 *                       the decision and, if the prediction does not hold (rejected step, or another damping), marks the
 *                       next tick as a rebuild-only tick instead of solving;
 *                       With <= 9 cameras the solve's launch also carries the back-substitution of the NEXT trial step (after a
 *                       speculative reduction, or inside mcba_lm_auto_tick on one GPU): the following mcba_lm_auto_trial /
 *                       mcba_lm_auto_tick then starts with the linearisation of that trial point.  MCBA_FUSE_BACKSUB=0 disables it.
 * mcba_lm_auto_tick   : trial + reduce + solve in one call; with a direct RCCL communicator attached (mcba_comm_init) the
 *                       library issues the collective(s) itself -- one per tick (speculative), or two with MCBA_SPECULATE=0;
 * mcba_lm_auto_wait   : spin on the ring slot until tick `seq` has posted (falls back to a stream synchronisation after
 *                       50 ms) and copy its MCBA_LM_STATE doubles.  At most 15 ticks may be outstanding.
 * After termination (state 15 != 0) the kernels of later ticks return at once; such ticks still post their slot. */
int mcba_lm_auto_config(mcba_handle* h, double ftol, double xtol, double gtol, double lam_min, double lam_max, const unsigned char* fixed_mask);
/* Floor of Nielsen's damping factor on an accepted step, lambda *= max(floor, 1 - (2 ratio - 1)^3); 0 (the default of a new handle)
 * = the classical 1/3.  Applies to every decision the library takes (mcba_lm_decide_reduce, mcba_lm_iterate, the device-resident
 * loop) and to the prediction its speculative Schur reduction is built on.  The Python driver sets 1/10 (solver.py). */
int mcba_lm_set_decrease_floor(mcba_handle* h, double dec_floor);
int mcba_lm_auto_solve(mcba_handle* h, unsigned long long seq, int decide);
int mcba_lm_auto_trial(mcba_handle* h, int decide);
int mcba_lm_auto_reduce(mcba_handle* h, int decide, int rank_slot);
int mcba_lm_auto_tick(mcba_handle* h, unsigned long long seq, int rank_slot);
int mcba_lm_auto_wait(mcba_handle* h, unsigned long long seq, double* state);
/* ---- whole stages of bundle_adjust() per crossing (round 5, ABI 6) ----------------------------------------------
 * mcba_lm_run: the device-resident loop from start to finish in ONE call -- what the caller otherwise drives through
 * mcba_set_params, mcba_linearize, mcba_reduce_fetch, mcba_lm_set_state, mcba_lm_auto_config, mcba_lm_auto_solve and then
 * mcba_lm_auto_tick / mcba_lm_auto_wait per iteration (the loop scipy's trf_no_bounds runs on the host: trf.py:401-560).  Nothing waits
 * between the upload of x0 and the first tick: the start state is written on the device from the reduced system of the start point,
 * the first `depth` ticks are enqueued behind the first solve, then the host polls the state ring.  The ticks, their order and every
 * decision are those of the per-call sequence (the device decides; the host only keeps `depth` ticks in flight).
 *   x0      12C + 6F doubles, or NULL to start from what parameter slot 0 holds (mcba_create_subset gathers it on the device)
 *   opt     13 doubles: 0 ftol 1 xtol 2 gtol (0 = that test off) 3 lambda_0 4 lambda_min 5 lambda_max 6 floor of Nielsen's factor
 *           (0 = 1/3) 7 curvature floor of the first linearisations (mcba_set_curvature_floor) 8 curvature switch fraction (state slot
 *           26; 0 = fixed model) 9 max_nfev 10 max iterations (< 0: none) 11 ticks in flight (1..12) 12 rank slot (0 on one GPU)
 *   fixed   n bytes (1 = camera-system variable held fixed) or NULL
 *   summary 4 doubles out: 0 status (scipy's 1..4; 0 = a limit was reached) 1 rows recorded 2 rows consumed by the loop proper (the
 *           rest were retired by the final drain of ticks still in flight) 3 iterations
 * MCBA_ERR_NONFINITE: "Residuals are not finite in the initial point." (least_squares.py).  With a direct RCCL communicator attached the
 * start system is all-reduced and every tick carries its collective, as in mcba_lm_auto_tick.
 * mcba_lm_history: the MCBA_LM_STATE doubles every retired tick posted, row 0 = the solve of the start point (cost, optimality).
 * mcba_lm_result: x of parameter slot `slot` (12C + 6F doubles -> x_out) and the gradient J^T f there (OptimizeResult.grad: the camera
 * gradient g_c of the reduce buffer scattered to the parameter layout, 0 where a parameter is held fixed, then the frame gradients),
 * packed on the device.  grad_out != NULL: x_out must have room for 2 (12C + 6F) doubles and grad_out == x_out + 12C + 6F -- one copy
 * brings both; grad_dev != NULL (and grad_out NULL): the gradient stays on the device as a mcba_buffer of its own (mcba_buffer_download /
 * mcba_buffer_free), for callers that attach it lazily; both NULL: x alone. */
int mcba_lm_run(mcba_handle* h, const double* x0, const double* opt, const unsigned char* fixed, double* summary);
int mcba_lm_history(mcba_handle* h, double* rows, size_t capacity_rows);
int mcba_lm_result(mcba_handle* h, int slot, double* x_out, double* grad_out, mcba_buffer** grad_dev);
/* k_solve_backsub's back-substitution workgroups wait for the solve of the same launch with a BOUNDED poll (~0.5 s).  If one ever
 * runs out, it stamps the tick's number into a device word and a host-mapped word: the next tick's decision discards its (stale)
 * trial point and only rebuilds the system, and mcba_lm_auto_wait switches the handle to the two-launch path (k_solve_cam, then
 * k_backsub) for good.  *timeouts = number of the last tick in which that happened (0: never), *fused = the fused launch is still
 * in use.  MCBA_FUSE_MAX_POLLS=0 forces the event (tests). */
int mcba_lm_fuse_status(mcba_handle* h, double* timeouts, int* fused);
/* The waiting side of that protocol.  Default since round 6 (on != 0): the readers ACQUIRE the release word with an agent-scope fence -- the
 * form the HIP memory model asks for.  on == 0: the readers poll and read with agent-scope relaxed atomic loads and rely on gfx950's in-order
 * issue -- measured ~1.3 us per iteration faster (1.3 % at 6 x 10 000 x 54), stress-tested bit-identical, but a data race under the HIP
 * memory model.  Chosen at run time per handle; a new handle takes MCBA_STRICT_SYNC from the environment (unset = 1; mcba_create_subset /
 * mcba_create_views inherit their source's setting).  Results are identical to the bit. */
int mcba_set_strict_sync(mcba_handle* h, int on);
int mcba_get_strict_sync(const mcba_handle* h);
/* The camera step (12C doubles) the last mcba_lm_auto_solve left on the device; synchronises. */
int mcba_get_cam_step(mcba_handle* h, double* host);
/* ---- direct RCCL (optional; frame-sharded runs) ------------------------------------------------------------
 * The library dlopen()s the RCCL already loaded in the process (torch's) -- it is not linked against it.
 * Rank 0 calls mcba_comm_unique_id (128 bytes), the caller broadcasts them (e.g. torch.distributed), every rank
 * calls mcba_comm_init; mcba_comm_allreduce then enqueues ncclAllReduce(SUM, f64, in place) on `count` doubles of the
 * reduce buffer starting at `offset`, on the handle's stream (no host synchronisation, no Python dispatch). */
int mcba_comm_unique_id(unsigned char* out128);
int mcba_comm_init(mcba_handle* h, const unsigned char* id128, int rank, int world);
int mcba_comm_allreduce(mcba_handle* h, size_t offset, size_t count);
int mcba_comm_count(mcba_handle* h, int* count);   /* ranks of the attached communicator (ncclCommCount); 0 if none */
int mcba_comm_destroy(mcba_handle* h);
/* Frame part of the gradient J^T f of the last mcba_build_reduced(): (F,6) doubles, host.
 * (The camera part is the g_c block of the reduce buffer.)  Feeds OptimizeResult.grad (trf.py:557-560). */
int mcba_get_frame_gradient(mcba_handle* h, double* host);

/* ---- robust triangulation (reference geometry.py:361-433 `triangulate`) ------------------------
 * Stateless: uvs (C, P, 2) detections (NaN = unseen), cam12 (C, 12) camera blocks in the parameter layout above,
 * dist5 (C, 5) OpenCV distortion (k1 k2 p1 p2 k3) or NULL (then k1, k2 of cam12), iterations of the undistortion
 * fixed point (OpenCV's default: 5).  out (P, 3): per-coordinate nan-median over all camera pairs of the linear (DLT)
 * two-view triangulations, NaN where fewer than two cameras see the point.  2 <= C <= 64 (up to 8 cameras: one lane per point,
 * everything in registers; beyond: one wavefront per point, the camera pairs across its lanes).  kernel_ms (may be NULL)
 * receives the kernel time measured with HIP events. */
int mcba_triangulate(int n_cameras, size_t n_points, const double* uvs, const double* cam12, const double* dist5, int iterations, int device, double* out, double* kernel_ms);

/* ---- the wrapper's frame pre-filter (bundle_adjustment.py:265-285) and frame subsets ------------------------ */
/* Reprojection error |observed - predicted| of every detection at x[slot].  Host outputs, both (C,F) row-major:
 * mean_cf = np.nanmean over the board points (NaN where the camera does not see the frame), full_cf = number of
 * points with both coordinates present (== N  <=>  the detection is complete: bundle_adjustment.py:266).  The per-point
 * errors stay on the GPU for mcba_error_median. */
int mcba_frame_errors(mcba_handle* h, int slot, double* mean_cf, double* full_cf);
/* np.nanmedian of those per-point errors over the frames with frame_mask[f] != 0 (F bytes; NULL = all frames): the exact
 * order statistic (radix select), mean of the two middle values for an even count; *count = number of values. */
int mcba_error_median(mcba_handle* h, const unsigned char* frame_mask, double* median, double* count);
/* One pass of that radix select for a caller that holds only a SHARD of the frames (frame-sharded bundle_adjust: every rank runs
 * the pre-filter on its own slice): hist256[b] = number of per-point errors of the frames with frame_mask[f] != 0 (F bytes; NULL =
 * the mask of the previous call) whose leading `pass` bytes equal `prefix` and whose next byte is b (pass 0 = most significant
 * byte of the IEEE bit pattern; the errors are non-negative, so bit-pattern order is numeric order).  The caller sums the
 * histograms over the ranks, picks the bin that holds the wanted rank, extends the prefix and calls again: 8 passes give the exact
 * order statistic whatever the sharding.  Synchronises. */
int mcba_error_histogram(mcba_handle* h, const unsigned char* frame_mask, unsigned long long prefix, int pass, unsigned long long* hist256);
/* The whole pre-filter (bundle_adjustment.py:265-285) in ONE call and one host synchronisation (round 5, ABI 6): upload of the
 * observations + board (both NULL: those already uploaded) and of x (12C + 6F: the parameters every frame is scored at, slot 0),
 * re-layout, the reprojection errors, and -- on the device -- which frames are complete in at least two cameras (:266), the worst
 * camera's nan-mean error per frame (:279), the threshold (outlier_threshold, or NaN: 5 x np.nanmedian of the used frames' per-point
 * errors, :281-282, exact by radix select) and the comparison (:285).  status: F bytes out, bit 0 = frame used (:266), bit 1 = excluded
 * as an outlier, bit 2 = complete in every camera.  info: 8 doubles out, 0 threshold 1 median 2 number of values under the median
 * 3 (1 = the median's candidate list overflowed and the eight-pass select of mcba_error_median ran instead) 4 frames used 5 excluded
 * 6 kept frames that are incomplete in some camera.  The per-point errors and per-(camera, frame) statistics stay on the device as
 * after mcba_frame_errors(h, 0, ...). */
int mcba_prefilter(mcba_handle* h, const double* uvs, const double* objpoints, const double* x, double outlier_threshold, unsigned char* status, double* info8);
/* mcba_prefilter + the gather of the frames it kept, when no random draw stands in between (bundle_adjustment.py:292-296 draws the subsample from the
 * caller's global numpy RNG only if n_frames <= the number of frames kept; n_frames < 0 = None).  info8[7]: 0 nothing kept, 1 the caller must draw
 * (*sub stays NULL: mcba_create_subset of its draw), 2 every frame kept, in order (solve on h itself), 3 *sub = a new handle holding the kept frames
 * in order (as mcba_create_subset makes it: observations, board and the parameters of slot 0 gathered on the device).  The caller owns *sub. */
int mcba_prefilter_subset(mcba_handle* h, const double* uvs, const double* objpoints, const double* x, double outlier_threshold, int n_frames, unsigned char* status, double* info8,
                          mcba_handle** sub);
/* New handle on the same device / stream holding the observations of n_frames frames of `src` (indices into its frames,
 * any order, repeats allowed) -- gathered device to device: what bundle_adjust solves on after the pre-filter, without
 * a second host upload (the reference slices all_calib_uvs[:, use_frames]: bundle_adjustment.py:298,312).  Parameter slot 0 of the
 * new handle = the camera blocks of src's slot 0 + the poses of the chosen frames (ABI 6).  Does not synchronise. */
int mcba_create_subset(mcba_handle** out, mcba_handle* src, const int* frames, int n_frames);

/* ---- calibrate() on the device (ABI 7, round 6) ------------------------------------------------------------------------
 * The initialiser that produces bundle_adjust()'s inputs: reference multicam_calibration/calibration.py:280-373.  Its two OpenCV calls per
 * view (cv2.calibrateCamera :68 on <= 100 sampled views per camera, cv2.solvePnP :108 on every complete view) and its pose graph
 * (:116-277) run on the detections a handle already holds (mcba_upload_observations).  A call of calibrate() is
 *   mcba_calib_complete -> [host: np.random.choice per camera, as the reference draws it :57-60] -> mcba_calib_start (the sampled views'
 *   homographies, Zhang's closed form for every camera's K from them, the views' poses with that K and no distortion: one crossing; its
 *   first and last step alone are mcba_calib_homographies and mcba_calib_view_poses)
 *   -> mcba_create_views + mcba_lm_run + mcba_lm_result (EVERY camera's fx fy cx cy k1 k2 and its views' poses in one device-resident LM run)
 *   -> mcba_calib_poses (every (camera, frame): one launch, the poses stay on the device) -> [host: maximum spanning tree of the C x C
 *   co-detection counts :146-197] -> mcba_calib_graph (the tree's pairwise medians, the chain of C - 1 transforms :230-235 and the consensus in
 *   one crossing; its pieces alone: mcba_calib_pairwise, mcba_calib_consensus).
 * Views are (camera, frame) int pairs.  intr9 = C x (fx fy cx cy k1 k2 p1 p2 k3) (OpenCV's five-coefficient model; the reference's defaults
 * leave p1 = p2 = k3 = 0).  Poses are board -> camera 6-vectors (rotation vector, translation), NaN rows where there is none.  The board must
 * be planar (z = 0), as the reference's chessboards are.  cv2 is absent from the build image: parity with OpenCV's numbers is unpinned; the
 * kernels are checked against numpy restatements (oracle/calibration_oracle.py) and against exact recovery of synthetic truth. */
/* complete_cf (C, F) bytes: 1 = all 2 N scalars of the detection are present (what :55 samples from and :107 solves). */
int mcba_calib_complete(mcba_handle* h, unsigned char* complete_cf);
/* Board-plane -> pixel homographies, H[2][2] = 1, of the listed views (Hartley-normalised DLT, the smallest singular vector of the 2N x 9
 * system): H_out n_views x 9 row-major (NaN for an incomplete view), ok_out n_views bytes or NULL. */
int mcba_calib_homographies(mcba_handle* h, const int* views, int n_views, double* H_out, unsigned char* ok_out);
/* cv2.solvePnP's job for the listed views: undistort (undistort_iterations rounds of OpenCV's fixed point), homography start, pose from it,
 * Levenberg-Marquardt on the pixel reprojection error (at most max_evaluations linearisations per view, every view its own damping).
 * poses_out n_views x 6, ok_out n_views bytes or NULL. */
int mcba_calib_view_poses(mcba_handle* h, const int* views, int n_views, const double* intr9, int undistort_iterations, int max_evaluations, double* poses_out, unsigned char* ok_out);
/* The closed-form start of cv2.calibrateCamera (:68) for EVERY camera in one crossing: homographies of the listed views; Zhang's camera matrix
 * per camera from its views (image of the absolute conic with the skew held at zero: the null vector of a 6-column system, found as the
 * smallest eigenvector of its 6 x 6 normal matrix by cyclic Jacobi; image_sizes = C x (width, height) in pixels sets the normalisation and
 * the fallback f = max(w, h), c = ((w - 1) / 2, (h - 1) / 2) for a camera with fewer than two usable views or a non-positive-definite
 * estimate); then mcba_calib_view_poses' work with that K and zero distortion.  k4_out C x (fx fy cx cy); closed_out C bytes or NULL (1 = the
 * closed form was used); poses_out n_views x 6 (NaN rows = none); ok_out n_views bytes or NULL. */
int mcba_calib_start(mcba_handle* h, const int* views, int n_views, const double* image_sizes, int undistort_iterations, int max_evaluations, double* k4_out, unsigned char* closed_out,
                     double* poses_out, unsigned char* ok_out);
/* estimate_pose (:74-113) of EVERY camera at once.  The poses stay on the device for the two calls below; poses_out (C,F,6), ok_out (C,F)
 * bytes, evals_out (C,F) bytes (linearisations a view took) are optional -- with all three NULL the call does not synchronise. */
int mcba_calib_poses(mcba_handle* h, const double* intr9, int undistort_iterations, int max_evaluations, double* poses_out, unsigned char* ok_out, unsigned char* evals_out);
/* estimate_pairwise_camera_transform (:116-143) for camera pairs edges = n_edges x (c1, c2): transforms_out (n_edges, 6) = component-wise
 * median over the frames both cameras have a pose for of T2 T1^-1 (exact order statistics: radix select on the device; NaN if the pair shares
 * no frame); counts_out (n_edges) = frames shared, or NULL. */
int mcba_calib_pairwise(mcba_handle* h, const int* edges, int n_edges, double* transforms_out, double* counts_out);
/* consensus_calib_poses (:239-277): extrinsics (C,6) world -> camera; poses_out (F,6) = per-coordinate nan-median over the cameras of
 * T_ext^-1 T_pose; NaN rows for frames no camera has a pose for. */
int mcba_calib_consensus(mcba_handle* h, const double* extrinsics, double* poses_out);
/* The pose graph of calibrate() (:200-277) in ONE crossing: mcba_calib_pairwise's medians for the spanning tree's edges, chained from `root` into
 * the world -> camera extrinsics ON THE DEVICE (:226-235), and mcba_calib_consensus with them.  edges = n_edges x (c1, c2) ordered so that c1 is
 * `root` or the c2 of an earlier edge and every camera is reached exactly once (the reference's tree sorted by distance from the root; n_edges =
 * C - 1, none for one camera).  extrinsics_out (C, 6), the root's row exactly 0; poses_out (F, 6); transforms_out (n_edges, 6) and counts_out
 * (n_edges) or NULL. */
int mcba_calib_graph(mcba_handle* h, const int* edges, int n_edges, int root, double* extrinsics_out, double* poses_out, double* transforms_out, double* counts_out);
/* A new handle of C cameras x n_views frames: frame j holds view j's detection in ITS camera alone (NaN in the others) -- the sampled views of
 * every camera side by side, so that one LM run (12 C camera parameters of which the extrinsics are held fixed at 0, 6 per view) is
 * get_intrinsics of every camera.  Gathered device to device; does not synchronise. */
int mcba_create_views(mcba_handle** out, mcba_handle* src, const int* views, int n_views);
/* The two pose-graph steps for a caller's own pose array (C,F,6) (NaN rows = none): the reference's public functions of the same names take
 * exactly that.  Stateless: host arrays in, host arrays out. */
int mcba_pose_pairwise(int n_cameras, int n_frames, const double* poses, const int* edges, int n_edges, int device, double* transforms_out, double* counts_out);
int mcba_pose_consensus(int n_cameras, int n_frames, const double* poses, const double* extrinsics, int device, double* poses_out);

/* ---- geometry helpers and diagnostics around the solver ---------------------------------------- */
/* undistort_points (geometry.py:328-358 = cv2.undistortPoints(uvs, K, dist, None, K)): n_points (u,v) pairs, K4 = (fx fy cx cy),
 * dist5 = (k1 k2 p1 p2 k3) or NULL; fixed-point iteration, `iterations` rounds (OpenCV's default: 5).  NaN rows stay NaN. */
int mcba_undistort_points(size_t n_points, const double* uvs, const double* K4, const double* dist5, int iterations, int device, double* out);
/* Single-camera calibration with OpenCV's FIVE-coefficient distortion model (k1 k2 p1 p2 k3) -- what the reference's get_intrinsics() asks of
 * cv2.calibrateCamera when fix_k3 / zero_tangent_dist are False (calibration.py:11-71) and of cv2.solvePnP when such coefficients come back
 * (:74-113).  Stateless: uvs (V,N,2) detections of V views (NaN = missing), objpoints (N,3), intr9 = fx fy cx cy k1 k2 p1 p2 k3, poses (V,6)
 * board -> camera (rotation vector, translation).  out (V,136) per view: the upper triangle (row by row, 120) of the Gauss-Newton block J^T J over
 * the parameters [intr9 | pose6], the gradient J^T r (15), the cost 0.5 sum r^2, with r = observed - predicted.  Rows are differentiated by
 * forward-mode automatic differentiation on the GPU (csrc/mcba_calib.hip); the LM iteration around it is calibration.py's. */
int mcba_calib_normal_equations(int n_views, int n_points, const double* uvs, const double* objpoints, const double* intr9, const double* poses, int device, double* out);
/* Numeric core of plot_residuals (viz.py:166-186) at x[slot]: per (camera, frame) with a complete detection the
 * least-squares homography from the undistorted detections to the board plane, the distortion-free reprojection of the
 * board mapped through it, and per camera the median distance to the board points (board units).
 * dist5: (C,5) distortion used for undistorting (NULL: (k1, k2, 0, 0, 0) of x[slot]).  Host outputs: median_error (C);
 * reprojections (C,F,N,2) and transformed (C,F,N,2, NaN where the detection is incomplete) -- either may be NULL. */
int mcba_reprojection_diagnostics(mcba_handle* h, int slot, const double* dist5, int undistort_iterations, double* median_error, double* reprojections, double* transformed);

/* ---- measurement ----------------------------------------------------------------------------- */
/* When enabled kernel launches are bracketed by hipEvents on the handle's stream.  `on` = 0: off; 1: every
 * kernel; otherwise a bit mask over the kernels in mcba_profile_names() order, shifted left by one
 * (bit k+1 selects kernel k) -- e.g. time only the dominant kernel inside a measured region. */
int mcba_profile_enable(mcba_handle* h, int on);
/* Bracket only every `stride`-th launch of each selected kernel (event records put barrier packets on the stream:
 * sampling keeps a measured region undisturbed).  Reset to 1 by mcba_profile_enable. */
int mcba_profile_stride(mcba_handle* h, int stride);
/* on != 0: the fused k_gram kernel is timed by events attached to its dispatch (the kernel's own begin / end timestamps -- the
 * figure rocprofv3 reports) instead of event records around the launch; other kernels and k_gram's other launch variants are
 * bracketed as before.  Reset by mcba_profile_enable. */
int mcba_profile_exact(mcba_handle* h, int on);
/* Microseconds an event bracket reads with nothing between its two records, mean of `pairs` (<= 4096) brackets on the handle's stream:
 * the bracket's own share of an event-timed kernel (bench.py subtracts it: `roofline.avg_launch_us`).  Synchronises. */
int mcba_profile_bracket_overhead(mcba_handle* h, int pairs, double* us);
/* Drains the recorded events.  names: '\n'-separated kernel names in the order of ms[] / calls[]. */
int mcba_profile_read(mcba_handle* h, double* ms_total, int* calls, int capacity, int* n_kernels);
const char* mcba_profile_names(void);
/* Measurement aid (round 4): TFLOP/s of independent v_fma_f64 (2 flop each) that `device` sustains with one wavefront per SIMD on every
 * compute unit -- the occupancy and the operand pattern of k_gram's accumulate block.  bench.py reports k_gram's FP64 instruction stream
 * against it next to the datasheet peak (the reference has no counterpart: its arithmetic is numpy / scipy on the host). */
int mcba_fp64_issue_rate(int device, double* tflops);
int mcba_synchronize(mcba_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MCBA_H */
