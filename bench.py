#!/usr/bin/env python
"""bench.py -- LM iterations/sec + ms per Jacobian-eval of the bundle-adjustment hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  * N = 1: runs in this process.
  * N > 1 under torch.distributed.run (WORLD_SIZE set): one rank per GPU, RCCL.
  * N > 1 WITHOUT a launcher (plain `python bench.py --gpus N ...`): this process starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD before it touches any GPU, relays
    rank 0's JSON line and exits with the children's status.  If the box shows fewer than N GPUs the ranks share
    GPU 0 and the collectives are host-staged over gloo (a REHEARSAL, flagged in the JSON; at most 6 ranks).

Workloads
  default (weak scaling; BASELINE.json metric / configs[2] per GPU): 6 cameras x 10 000 frames x 54 points PER GPU, full
      intrinsics + distortion + extrinsics + poses, soft-L1, synthetic board detections (multicam_calibration_amd.synth,
      seed 0).  `value` = steps x n_gpus / time: LM iterations/sec in units of one 6x10kx54 problem -- at N = 1 exactly
      BASELINE's "LM iterations/sec" on config 3.
  --frames-total T (strong scaling; BASELINE configs[3] with T = 100000): T frames of ONE rig sharded contiguously over
      the N ranks (12 500 per rank at N = 8).  `value` = LM iterations/sec of the whole T-frame problem.

A "step" is ONE complete Levenberg-Marquardt iteration on the rank's shard (a "tick" of the device-resident loop):
back-substitution (k_backsub), residuals + analytic Jacobian blocks + normal equations of the trial point in one pass over
the observations (k_gram; its cost decides accept/reject), trial sums + decision + frame factors + Schur product (k_syrk),
k_reduce_system, [all-reduce], the (12C)^2 reduced camera solve (k_solve_cam).  Every step linearises its trial
point, accepted or not (never less work than a real iteration).

`value` is the MEDIAN of `--windows` (default 5) timed regions of exactly `--steps` steps each -- every region bracketed by barrier +
synchronize on both sides and reduced by MAX over the ranks; all of them are printed (`value_windows`).

Extra objects on the same line: `roofline` (dominant kernel of the timed region, HIP-event timed on the launch stream;
`bound` names the binding roof), `jacobian_eval` (BASELINE's second figure: ms per materialised Jacobian-eval, with its own
HBM roofline), `cpu_baseline` (the oracle's scipy path on a bounded sample of the same workload, rank 0 at N = 1 only), `configs` (all
five BASELINE configs, per iteration), `end_to_end` / `end_to_end_other_shapes` (wall time of the calls users make: `bundle_adjust()`,
`calibrate()` -- the call that produces its inputs --, the pipeline of the two, the off-default solver paths), `strict_sync` (what the
relaxed reader of the solve's release word would save against the acquiring one that is the default).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
# scripts/micro/write_bw.hip on the gpurun MI355X (1 GiB, 16 B/lane): streaming write 4.5-5.9 TB/s, copy 4.8-5.5, read 6.2-6.6
HBM_MEASURED_WRITE_GBS = 5930.0
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (AMD datasheet; 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
FP64_VALU_MEASURED_TFLOPS = 57.0  # scripts/micro/fma_f64_rate.hip on the same GPU: 54-58 TFLOP/s of independent v_fma_f64 (one per 4.82 clock64 ticks)

C, F_PER_GPU, ROWS, COLS = 6, 10000, 6, 9
# FP64 instruction mix of k_gram: profiles/gram_flops.json, written by scripts/gram_isa_mix.py -- the 4-point loop body of the
# instance this workload runs (k_gram<soft_l1, planar + unit f_scale>) counted in the ISA of the built kernel (FMA / MUL / ADD /
# other FP64 per point-observation), the per-(camera, frame) remainder taken from the SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 counters of
# the profile named in that file.  REAL flops = 2 FMA + MUL + ADD; issue slots = every FP64 instruction (priced as an FMA).
GRAM_FLOPS_FILE = os.path.join(ROOT, "profiles", "gram_flops.json")
# algorithmic bytes of one fused LM iteration at 6 x 10 000 x 54 (observations read once for the trial
# linearisation, W / V / g_f written, re-read by the Schur pass and by the back-substitution)
TICK_ALGORITHMIC_BYTES_10K = 218e6


def algorithmic_bytes(kernel, C, F, N):
    """Bytes one launch must move if every operand is read / written exactly once."""
    n = 12 * C
    obs = 16 * C * F * N
    poses = 48 * F
    rec = 800 * C * F
    return {
        "k_gram": obs + poses + rec + 736 * C * ((F + 63) // 64),
        "k_cost": obs + poses,
        "k_syrk": 576 * C * F + 216 * C * F + 320 * F,
        "k_backsub": 576 * C * F + 320 * F + 2 * poses,
        "k_jacobian": obs + poses + (288 + 16) * C * F * N,
        "k_reduce_system": 8 * (n * n + 3 * n),
        "k_sum_trial": 64,
    }[kernel]


def gram_work(C, F, N):
    """(real FP64 flops, FP64 issue slots x 2, per-point mix) of one k_gram launch.  Every lane of every wavefront runs the
    point loop, the lanes of the padding frames included (Fpad = frames rounded up to 64): that is what the counters see."""
    with open(GRAM_FLOPS_FILE) as fh:
        g = json.load(fh)
    pp, out = g["per_point_observation"], g["per_pair_outside_loop"]
    lanes = C * ((F + 63) // 64) * 64
    flop_pt = 2 * pp["fma"] + pp["mul"] + pp["add"]
    real = lanes * (N * flop_pt + out["flop"])
    slots = 2 * lanes * (N * pp["fp64"] + out["fp64"])
    return real, slots, {"fma": pp["fma"], "mul": pp["mul"], "add": pp["add"], "other_fp64": pp["other_f64"], "fp64": pp["fp64"], "accvgpr_moves": pp["accvgpr_mov"],
                         "other_valu": pp["valu_other"], "valu": pp["valu"], "fp64_outside_loop_per_pair": out["fp64"], "source": "profiles/gram_flops.json (" + g["source"] + ")"}


def end_to_end(m, p, reps=5, full=True):
    """The call users make (reference bundle_adjustment.py:195 -> 5-tuple): host arrays in, 5-tuple out, return_jac=False, default
    tolerances, warm.  Wall-clock per stage: every ops.Problem method is a C-ABI crossing (those that return host data
    synchronise), the host-side stages are timed around them."""
    import contextlib
    import functools
    import io

    from multicam_calibration_amd import api, ops, solver

    acc = {}

    def timed(name, fn):
        @functools.wraps(fn)
        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return w

    saved = []
    for owner, names in ((ops.Problem, ["__init__", "prefilter", "prefilter_subset", "subset", "close", "lm_run", "lm_result", "residuals_detach", "set_x_scale"]),
                         (api, ["select_frames", "deserialize_params", "serialize_params"]), (solver, ["lm_solve"])):
        for n in names:
            saved.append((owner, n, getattr(owner, n)))
            setattr(owner, n, timed(("ops." if owner is ops.Problem else owner.__name__.split(".")[-1] + ".") + n, getattr(owner, n)))
    Cc, F, Nn = p["uvs"].shape[:3]

    def run():
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            return m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=F, verbose=0, return_jac=False)

    try:
        run()
        run()
        acc.clear()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = run()
            times.append(1e3 * (time.perf_counter() - t0))
        res = out[4]
        t_fun0 = time.perf_counter()
        nres = int(res.fun.size)   # first read of the lazily attached residual vector: the device-to-host copy happens here
        t_fun = 1e3 * (time.perf_counter() - t_fun0)
        acc_main = dict(acc)   # (the stage timers keep running below: the breakdown is of the `reps` calls above only)
        # the reference's own call, no extra keyword at all (return_jac defaults to True): `result.jac` is lazy too, the result
        # keeps the GPU handle until it is dropped
        dflt = []
        for _ in range(3 if full else 0):
            np.random.seed(0)
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                o = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=F, verbose=0)
            dflt.append(1e3 * (time.perf_counter() - t0))
            del o
        # a real recording: 5 % of the (camera, frame) detections missing -> the lazy fields need their row mask at call time
        # (mcba_seen_bits: taken from the GPU's copy of the observations)
        uvs_m = p["uvs"].copy()
        uvs_m[np.random.default_rng(7).random(uvs_m.shape[:2]) < 0.05] = np.nan
        miss = []
        for _ in range(4 if full else 3):
            np.random.seed(0)
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                o = m.bundle_adjust(uvs_m, p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=F, verbose=0, return_jac=False)
            miss.append(1e3 * (time.perf_counter() - t0))
            del o
        # n_frames larger than the recording (the reference's default n_frames=10000 on anything shorter; its tutorial passes 5000 for 2 130 frames):
        # no random draw stands between the selection and its gather (bundle_adjustment.py:292-296) -- every frame kept = no gather at all
        nodraw = []
        for _ in range(4 if full else 3):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                o = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=2 * F, verbose=0, return_jac=False)
            nodraw.append(1e3 * (time.perf_counter() - t0))
            del o
        nodraw_m = []
        for _ in range(4 if full else 3):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                o = m.bundle_adjust(uvs_m, p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=2 * F, verbose=0, return_jac=False)
            nodraw_m.append(1e3 * (time.perf_counter() - t0))
            del o
    finally:
        for owner, n, f in saved:
            setattr(owner, n, f)
    ms = {k: 1e3 * v / reps for k, v in acc_main.items()}
    total = float(np.median(times))
    pre = ms.get("ops.prefilter", 0.0) + ms.get("ops.prefilter_subset", 0.0) + ms.get("ops.__init__", 0.0)
    sel = ms.get("api.select_frames", 0.0) - pre
    lm = ms.get("solver.lm_solve", 0.0)
    gather = ms.get("ops.subset", 0.0)
    close = ms.get("ops.close", 0.0)
    detach = ms.get("ops.residuals_detach", 0.0)
    out = {"ms": total, "ms_min": float(min(times)), "reps": reps, "nfev": int(res.nfev), "status": int(res.status),
           "what": f"bundle_adjust(host arrays ({Cc},{F},{Nn},2) -> 5-tuple), n_frames={F}, return_jac=False, default tolerances (ftol=1e-4), warm (third call onwards), median of {reps}",
           "breakdown_ms": {"prefilter_one_crossing_h2d_relayout_scores_selection": pre, "host_selection_print_rng": sel, "device_gather_of_selection": gather,
                            "lm_solve_total": lm, "of_which_lm_run_one_crossing": ms.get("ops.lm_run", 0.0), "of_which_lm_result_d2h": ms.get("ops.lm_result", 0.0),
                            "residual_vector_left_on_device": detach, "handle_teardown": close, "python_rest": total - (pre + sel + gather + lm + detach + close)},
           "result_fun_first_read_ms": t_fun, "result_fun_size": nres, "missing_detections_call_ms": float(np.median(miss[1:])),
           "n_frames_above_recording_call_ms": float(np.median(nodraw[1:])), "n_frames_above_recording_missing_detections_call_ms": float(np.median(nodraw_m[1:])),
           "note": "three C-ABI crossings carry the call (mcba_prefilter, mcba_lm_run, mcba_lm_result); result.fun / result.grad stay on the GPU until first read (LazyOptimizeResult) -- the download of fun is timed separately above and is not part of `ms`; missing_detections_call_ms = the same call with 5 % of the (camera, frame) detections NaN; n_frames_above_recording_call_ms = the same call with n_frames larger than the recording (the reference's default 10000 on a shorter recording; its tutorial's 5000 for 2 130 frames): no random draw, the kept frames are gathered inside the pre-filter's crossing (mcba_prefilter_subset) or, all kept, not at all"}
    if full:
        out["default_call_ms"] = float(np.median(dflt))
        out["note"] += "; default_call_ms = the same call without return_jac=False (result.jac lazy, the result holds the handle)"
    return out


def calibrate_timing(m, p, reps=5):
    """calibrate() -- the call that produces bundle_adjust()'s inputs (reference calibration.py:280-373; its tutorial: 60 s of intrinsics + 2.7 s of
    PnP for 6 cameras x 2 130 frames) -- warm, median of `reps`, with its stages, and the pipeline a user runs: calibrate() -> bundle_adjust()."""
    import contextlib
    import functools
    import io

    from multicam_calibration_amd import calibration as cal, ops

    acc = {}

    def timed(name, fn):
        @functools.wraps(fn)
        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return w

    stages = {"upload": [(ops.Problem, "__init__"), (ops.Problem, "calib_complete")],
              "intrinsics_sampled_views": [(cal, "_sample_all_cameras"), (cal, "_start_on_device"), (cal, "_refine_intrinsics_on_device")],
              "poses_every_view": [(ops.Problem, "calib_poses")],
              "pose_graph_and_consensus": [(cal, "_pose_graph_on_device")]}
    extra = [(ops.Problem, "calib_start"), (ops.Problem, "lm_run"), (ops.Problem, "calib_graph")]
    saved = []
    for owner, n in [x for v in stages.values() for x in v] + extra:
        saved.append((owner, n, getattr(owner, n)))
        setattr(owner, n, timed(n, getattr(owner, n)))
    Cc, F, Nn = p["uvs"].shape[:3]
    sizes = [(1280, 1024)] * Cc

    def run():
        np.random.seed(0)
        return cal.calibrate(p["uvs"], sizes, p["obj"], verbose=False)

    try:
        run()
        run()
        times, per_run = [], []
        for _ in range(reps):
            acc.clear()
            t0 = time.perf_counter()
            ext, intr, poses, tree = run()
            times.append(1e3 * (time.perf_counter() - t0))
            per_run.append(dict(acc))
    finally:
        for owner, n, f in saved:
            setattr(owner, n, f)
    acc_main = {k: reps * float(np.median([r.get(k, 0.0) for r in per_run])) for k in set().union(*per_run)}   # (medians per stage, like the total: one slow run does not tilt the table)
    ok = ~np.isnan(poses).any(1)
    uv_ok = p["uvs"] if ok.all() else np.ascontiguousarray(p["uvs"][:, ok])   # (frames no camera saw have no start pose)
    poses_ok = poses[ok]
    ba = []
    for _ in range(reps + 1):
        np.random.seed(0)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = m.bundle_adjust(uv_ok, ext, intr, p["obj"], poses_ok, n_frames=int(ok.sum()), verbose=0, return_jac=False)
        ba.append(1e3 * (time.perf_counter() - t0))
    ms = {k: 1e3 * v / reps for k, v in acc_main.items()}
    total = float(np.median(times))
    br = {name: sum(ms.get(n, 0.0) for _, n in members) for name, members in stages.items()}
    br["python_rest"] = total - sum(br.values())
    br["of_which"] = {"closed_form_start_crossing_homographies_zhang_view_poses": ms.get("calib_start", 0.0), "joint_refinement_lm_run": ms.get("lm_run", 0.0),
                      "pose_graph_crossing_medians_chain_consensus": ms.get("calib_graph", 0.0)}
    ba_ms = float(np.median(ba[1:]))
    return {"calibrate_ms": total, "calibrate_ms_min": float(min(times)), "reps": reps, "breakdown_ms": br, "bundle_adjust_from_calibrate_ms": ba_ms, "pipeline_ms": total + ba_ms,
            "bundle_adjust_from_calibrate": {"nfev": int(out[4].nfev), "status": int(out[4].status), "cost": float(out[4].cost), "frames": int(ok.sum())},
            "what": f"calibrate(host array ({Cc},{F},{Nn},2), image sizes, board) -> (extrinsics, intrinsics, consensus poses, spanning tree), n_samples_for_intrinsics=100, warm, median of {reps}; "
                    "then bundle_adjust() on its outputs (n_frames = all, return_jac=False, default tolerances); pipeline_ms = the sum",
            "reference": "docs/source/calibration_tutorial.ipynb:89,103: 60 s (intrinsics, 6 cameras) + 2.7 s (PnP) with OpenCV at 6 x 2 130 x 35; round 5 of this repository (host numpy starts): 478 ms / 9 888 ms at 6 x 2 130 x 35 / 6 x 10 000 x 54 on the same GPU box (profiles/round6/calibrate_baseline_r5.json)"}


def off_default_calls(m, p, reps=3):
    """The two off-default solver paths of the user-level call, timed: `bounds=` (k2 >= 0 on two cameras + a box on 20 poses; the reference's
    bounded solver is scipy's trf_bounds, bundle_adjustment.py:301-313) and a callable `loss=` (soft_l1 written as a function)."""
    import contextlib
    import io

    Cc, F, Nn = p["uvs"].shape[:3]
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    lo, hi = np.full(x0.size, -np.inf), np.full(x0.size, np.inf)
    for c in (1, 3 % Cc):
        lo[12 * c + 5] = 0.0   # k2 >= 0
    rng = np.random.default_rng(11)
    for f in rng.choice(F, min(20, F), replace=False):
        sl = slice(12 * Cc + 6 * f, 12 * Cc + 6 * f + 6)
        lo[sl], hi[sl] = x0[sl] - np.r_[0.002, 0.002, 0.002, 1.0, 1.0, 1.0], x0[sl] + np.r_[0.002, 0.002, 0.002, 1.0, 1.0, 1.0]
    x_in = np.clip(x0, lo, hi)

    inside = [0.0, 0]

    def soft_l1(z):
        t0 = time.perf_counter()
        t = 1.0 + z
        r = np.empty((3,) + z.shape)
        np.sqrt(t, out=r[1])
        r[0] = 2.0 * (r[1] - 1.0)
        np.divide(1.0, r[1], out=r[1])
        r[2] = -0.5 * r[1] / t
        inside[0] += time.perf_counter() - t0
        inside[1] += 1
        return r

    def call(**kw):
        ts, res = [], None
        for _ in range(reps + 1):
            np.random.seed(0)
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=2 * F, verbose=0, return_jac=False, **kw)[4]
            ts.append(1e3 * (time.perf_counter() - t0))
        return float(np.median(ts[1:])), res

    t_plain, r_plain = call()
    t_b, r_b = call(bounds=(lo, hi))
    inside[:] = [0.0, 0]
    t_c, r_c = call(loss=soft_l1)
    fn_ms = 1e3 * inside[0] / (reps + 1)
    return {"callable_loss_ms_inside_the_callers_function": fn_ms, "callable_loss_calls_of_the_function": inside[1] / (reps + 1), "callable_loss_ms_outside_the_function": t_c - fn_ms,"unbounded_call_ms": t_plain, "unbounded_nfev": int(r_plain.nfev),
            "bounds_call_ms": t_b, "bounds_nfev": int(r_b.nfev), "bounds_active": int(np.count_nonzero(r_b.active_mask)), "bounds_ms_per_evaluation": t_b / max(int(r_b.nfev), 1),
            "callable_loss_call_ms": t_c, "callable_loss_nfev": int(r_c.nfev), "callable_loss_ms_per_evaluation": t_c / max(int(r_c.nfev), 1),
            "what": f"bundle_adjust at ({Cc},{F},{Nn},2), n_frames above the recording, return_jac=False, default tolerances, warm, median of {reps}: plain; bounds = k2 >= 0 on two cameras + a box "
                    "(+-0.002 rad, +-1 mm) around the start pose of 20 frames; loss = soft_l1 as a Python callable (scipy's rho(z) -> (rho, rho', rho'') contract)"}


def cpu_baseline(sample_frames=1000, max_nfev=12):
    """the oracle's CPU path (vectorised numpy residual + the reference's own scipy.least_squares call:
    trf, soft_l1, x_scale='jac', ftol=1e-4, 2-point finite differences through jac_sparsity) on a bounded sample of the
    same workload, plus one timed `approx_derivative(..., sparsity=(A, groups))` = the CPU "Jacobian-eval", with the
    observed core utilisation (os.times / wall)."""
    from oracle import ba_oracle as orc
    from multicam_calibration_amd import synth
    from scipy.optimize import least_squares
    from scipy.optimize._numdiff import approx_derivative, group_columns

    # One thread: the path is sequential (sparse finite differences, LSMR); left alone, the BLAS / OpenMP pools behind numpy spin
    # on every core of the host (round 2 reported "14.9 cores busy" of 256 for what the survey measured as 1.6-1.8 cores of work).
    limiter = None
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=1)
    except Exception:  # noqa: BLE001 -- threadpoolctl missing: report what is observed
        pass
    p = synth.make_problem(C, sample_frames, rows=ROWS, cols=COLS, seed=0)
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    t0 = time.perf_counter()
    A = orc.sparsity_csr(p["uvs"])
    t_pat = time.perf_counter() - t0
    groups = group_columns(A)
    fun = lambda x: orc.residuals(x, p["uvs"], p["obj"])
    t0 = time.perf_counter()
    approx_derivative(fun, x0, method="2-point", sparsity=(A, groups))
    t_jac = time.perf_counter() - t0
    c0, t0 = os.times(), time.perf_counter()
    res = least_squares(orc.residuals, x0, jac_sparsity=A, verbose=0, x_scale="jac", ftol=1e-4, method="trf", loss="soft_l1", max_nfev=max_nfev, args=(p["uvs"], p["obj"]))
    dt = time.perf_counter() - t0
    c1 = os.times()
    cores = ((c1.user - c0.user) + (c1.system - c0.system)) / dt
    if limiter is not None:
        limiter.restore_original_limits()
    iters = max(res.njev - 1, 1)  # accepted iterations = Jacobian evaluations after the first
    it_per_s_sample = iters / dt
    scale = sample_frames / F_PER_GPU
    return {
        "value": it_per_s_sample * scale, "unit": "it/s", "cores": 1 if limiter is not None else round(cores, 2), "kind": "port",
        "config": {"cameras": C, "frames": sample_frames, "points": ROWS * COLS, "trf_iterations": iters, "nfev": int(res.nfev), "njev": int(res.njev), "seconds": dt,
                   "it_per_s_on_sample": it_per_s_sample, "scale_to_workload": scale, "scaling_rule": "cost per iteration linear in frames (BASELINE.md section 2)",
                   "host_cpu_count": os.cpu_count(), "cores_busy_observed": round(cores, 2), "threads": 1 if limiter is not None else None,
                   "cores_note": "thread pools limited to 1 (threadpoolctl); cores_busy_observed = os.times / wall of the least_squares call"},
        "sample": f"{C}x{sample_frames}x{ROWS * COLS} sample of the workload, {iters} TRF iterations (nfev {res.nfev}, njev {res.njev}) in {dt:.1f}s = {it_per_s_sample:.3f} it/s on the sample; "
                  f"scaled x{scale:g} to 10k frames (cost per iteration is linear in frames: BASELINE.md section 2); sparsity pattern {t_pat:.2f}s + colouring excluded; "
                  f"observed {cores:.2f} cores busy of os.cpu_count()={os.cpu_count()}",
        "ms_per_jacobian_eval_sample": 1e3 * t_jac,
        "ms_per_jacobian_eval_scaled_to_10k_frames": 1e3 * t_jac / scale,
        "jacobian_eval": f"one scipy approx_derivative(2-point, sparsity=(A, {int(groups.max()) + 1} colour groups)) of the oracle residuals on the sample",
    }


def tick_bytes(C, F, N, cw=12):
    """SURVEY 8(d)'s accounting of one fused LM iteration, for any shape: the trial linearisation reads the observations and poses
    and writes V_f / g_f per frame and the W block (cw x 6) per (camera, frame); the Schur pass and the back-substitution re-read
    them; the trial cost needs the observations once more.  218.8 MB at 6 x 10 000 x 54, cw = 12."""
    obs, poses, vg, w = 16 * C * F * N, 48 * F, 336 * F, 48 * cw * C * F
    return (obs + poses) + (vg + w) + (vg + w) + (vg + w + poses) + (obs + poses)


def other_configs(m, headline_ms_per_step):
    """LM-iteration time of the BASELINE configs that are not the bench workload, same loop, same counting (every iteration
    linearises its trial point), untimed with respect to `value`: configs[0] 2 x 50 x 54 (the reference's CPU-runnable case),
    configs[1] 6 x 1 000 x 54 with the intrinsics held fixed (the library's 6-wide camera block).  configs[2] is the headline."""
    out = {}
    for key, (Cc, Fc, fixed, rows, cols) in (("configs[0]", (2, 50, False, ROWS, COLS)), ("configs[1]", (6, 1000, True, ROWS, COLS)),
                                             ("configs[3] (one of 8 frame shards)", (6, 12500, False, ROWS, COLS)), ("configs[4] (one of 8 frame shards)", (24, 6250, False, 10, 20))):
        p = m.synth.make_problem(Cc, Fc, rows=rows, cols=cols, seed=0)
        x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
        prob = m.ops.Problem(p["uvs"], p["obj"])
        if fixed:
            assert prob.set_camera_block(6)
        lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0)
        lm.start(x0)
        big = Cc * Fc * rows * cols > 5_000_000
        for _ in range(40 if big else 100):
            lm.iterate(always_linearize=True)
        prob.synchronize()
        best = 1e9
        for _ in range(3):
            n0, t0 = lm.nfev, time.perf_counter()
            for _ in range(100 if big else 200):
                lm.iterate(always_linearize=True)
            lm.finalize()
            prob.synchronize()
            best = min(best, (time.perf_counter() - t0) / max(lm.nfev - n0, 1))
        prob.profile_enable(True)
        for _ in range(40):
            lm.iterate(always_linearize=True)
        kern = {k: round(1e3 * ms / n, 2) for k, (ms, n) in prob.profile_read().items() if n}
        prob.profile_enable(False)
        tb = tick_bytes(Cc, Fc, rows * cols, 6 if fixed else 12)
        out[key] = {"shape": f"{Cc} cameras x {Fc} frames x {rows * cols} points" + (", intrinsics held fixed (camera block 6 wide)" if fixed else ", all parameters free"),
                    "us_per_iteration": round(best * 1e6, 2), "it_per_s": round(1.0 / best, 1), "kernels_us_by_hip_events": kern,
                    "tick_roofline": {"bound": "hbm", "algorithmic_bytes_per_iteration": tb, "achieved": tb / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tb / best / 1e9 / HBM_PEAK_GBS},
                    "cost_end": lm.cost}
        prob.close()
        del p, prob, lm
    tb = tick_bytes(C, F_PER_GPU, ROWS * COLS)
    out["configs[2]"] = {"shape": f"{C} cameras x {F_PER_GPU} frames x {ROWS * COLS} points, all parameters free (the bench workload: `value`)", "us_per_iteration": round(1e3 * headline_ms_per_step, 2),
                         "it_per_s": round(1e3 / headline_ms_per_step, 1),
                         "tick_roofline": {"bound": "hbm", "algorithmic_bytes_per_iteration": tb, "achieved": tb / (headline_ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": tb / (headline_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    out["note"] = ("wall clock per LM iteration of the device-resident loop (best of 3 x 200 iterations after 100 warm-up iterations; 3 x 100 after 40 for the shards of configs[3] / [4]); "
                   "bytes = SURVEY 8(d)'s accounting generalised (bench.py: tick_bytes); configs[3] / [4] are ONE of the eight frame shards north_star partitions them into, on one GPU, without the collective")
    return out


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the ranks as CHILD processes (this process never touches a GPU --
    a process that has initialised the GPU must not exec or be replaced), relay their stdout, exit with their status."""
    import torch  # device_count() does not initialise the GPU on this image

    ndev = torch.cuda.device_count()
    env = os.environ.copy()
    if ndev < args.gpus:
        if args.gpus > 6:
            raise SystemExit(f"--gpus {args.gpus}: only {ndev} GPU(s) visible and a shared-GPU rehearsal is limited to 6 ranks")
        env["MCBA_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5, help="timed regions of --steps steps each; `value` = the median window (all are printed)")
    ap.add_argument("--frames", type=int, default=F_PER_GPU, help="frames per GPU, weak scaling (default = BASELINE config 3 per GPU)")
    ap.add_argument("--frames-total", type=int, default=0, help="strong scaling: this many frames of ONE rig sharded over the ranks (100000 = BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the wall-clock measurement of the user-level bundle_adjust() call")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the per-iteration times of BASELINE configs[0] and configs[1]")
    ap.add_argument("--prewarm", type=int, default=300, help="untimed clock-ramp iterations of a throw-away solve before the W warm-up steps")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # Libraries (RCCL prints a version banner) may write to stdout: keep fd 1 for the ONE JSON line only.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import multicam_calibration_amd as m

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("MCBA_BENCH_SHARE_GPU") == "1"  # rehearsal of N > 1 ranks on a one-GPU box (all ranks on device 0)
    if share_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    comm = None
    force_dist = os.environ.get("MCBA_BENCH_FORCE_DIST") == "1"  # exercise the RCCL plumbing with a single rank
    backend = "none"
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        torch.cuda.set_device(local_rank)
        if share_gpu:
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        comm = "pending"
    else:
        torch.cuda.set_device(0)
        local_rank = 0

    N = ROWS * COLS
    if args.frames_total:
        F = len(np.array_split(np.arange(args.frames_total), world)[rank])  # contiguous shards of one rig
        frames_total, scaling = args.frames_total, "strong"
    else:
        F = args.frames
        frames_total, scaling = F * world, "weak"
    p = m.synth.make_problem(C, F, rows=ROWS, cols=COLS, seed=0, frame_seed=rank if world > 1 else None)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = m.ops.Problem(p["uvs"], p["obj"], device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
    if comm is not None:
        comm = m.solver.make_comm(prob, torch.device(f"cuda:{local_rank}"))  # direct RCCL, else torch.distributed (gloo: host-staged)

    # Clock ramp (untimed, not part of the W warm-up steps): a fresh box starts at idle clocks and needs some tens of
    # milliseconds of work before the SMU settles; a throw-away solve of the same problem provides it, then the measured
    # solve starts again from x0.  (With --steps 20 --warmup 5 the whole measured run is 3 ms long.)
    pre = m.solver.LevenbergMarquardt(prob, comm, ftol=0.0, xtol=0.0, gtol=0.0)
    pre.start(x0)
    for _ in range(args.prewarm):
        pre.iterate(always_linearize=True)
    pre.finalize()
    prob.synchronize()

    lm = m.solver.LevenbergMarquardt(prob, comm, ftol=0.0, xtol=0.0, gtol=0.0)
    lm.start(x0)
    cost0 = lm.cost
    for _ in range(args.warmup):
        lm.iterate(always_linearize=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events on the launch stream bracket ONLY the dominant kernel inside the timed region (bracketing all
    # kernels of a step costs ~60 us of host time per step); the per-kernel table comes from an untimed pass.
    DOMINANT = "k_gram"
    prob.profile_enable(True, only=[DOMINANT], stride=8)  # every 8th launch: the event records must not pace the stream
    prob.profile_read()
    # `--windows` timed regions of EXACTLY `--steps` steps each, every one bracketed by barrier + synchronize on both sides and reduced by MAX
    # over the ranks; `value` comes from the MEDIAN window, all of them are printed (a single 2 ms window at the driver's --steps 20 is one
    # sample of a box's clock state, not a measurement)
    window_dts, ticks = [], 0
    for _w in range(max(1, args.windows)):
        barrier()
        nfev0 = lm.nfev
        t0 = time.perf_counter()
        wticks = 0
        while lm.nfev - nfev0 < args.steps:
            # a step = an LM iteration that evaluated (and linearised) a trial point.  Frame-sharded runs with one collective
            # per iteration occasionally need an extra rebuild-only pass (mispredicted damping): it is timed, not counted.
            status = lm.iterate(always_linearize=True)
            wticks += 1
            assert status is None, f"the LM loop stopped inside the timed region (status {status})"
            assert wticks <= 2 * args.steps, "too many rebuild-only passes"
        barrier()
        dtw = time.perf_counter() - t0
        assert lm.nfev - nfev0 == args.steps, f"{lm.nfev - nfev0} trial evaluations in {args.steps} timed steps"
        if dist is not None:
            tmax = torch.tensor([dtw], dtype=torch.float64, device="cpu" if backend == "gloo" else f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dtw = float(tmax.item())
        window_dts.append(dtw)
        ticks += wticks
    dt = float(np.median(window_dts))
    prof_timed = prob.profile_read()
    prob.profile_enable(True)
    for _ in range(min(args.steps, 50)):
        lm.iterate(always_linearize=True)
    prof = prob.profile_read()
    # the dominant kernel alone, every launch bracketed, 50 iterations (>= 25 samples whatever --steps is), and what an EMPTY bracket reads
    # on the same stream: the event pair's own share of each sample
    prob.profile_enable(True, only=[DOMINANT])
    for _ in range(50):
        lm.iterate(always_linearize=True)
    prof_dom = prob.profile_read()[DOMINANT]
    dom_events_us = 1e3 * prof_dom[0] / max(prof_dom[1], 1)
    # ... and once more with the events attached to the kernel's DISPATCH (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps,
    # the figure rocprofv3 reports -- an event pair recorded around a launch also reads the packet latencies on either side of it)
    prob.profile_enable(True, only=[DOMINANT], exact=True)
    for _ in range(50):
        lm.iterate(always_linearize=True)
    prof_exact = prob.profile_read()[DOMINANT]
    prob.profile_enable(False)
    bracket_us = prob.profile_bracket_overhead(200)
    # what the relaxed reader of the solve's release word would save (MCBA_STRICT_SYNC=0 / mcba_set_strict_sync(h, 0): relaxed agent-scope loads
    # relying on gfx950's in-order issue instead of the acquire fence that is the default since round 6): alternating blocks of 200 iterations
    strict = None
    if hasattr(prob, "set_strict_sync") and dist is None:
        default_mode = bool(prob.lib.mcba_get_strict_sync(prob.handle))
        per = {False: [], True: []}
        for rep in range(3):
            for mode in (True, False):
                prob.set_strict_sync(mode)
                for _ in range(10):
                    lm.iterate(always_linearize=True)
                lm.finalize()
                prob.synchronize()
                n0, t0s = lm.nfev, time.perf_counter()
                for _ in range(200):
                    lm.iterate(always_linearize=True)
                lm.finalize()
                prob.synchronize()
                per[mode].append(1e6 * (time.perf_counter() - t0s) / max(lm.nfev - n0, 1))
        prob.set_strict_sync(default_mode)
        strict = {"default": "acquire" if default_mode else "relaxed", "acquire_us_per_iteration": min(per[True]), "relaxed_us_per_iteration": min(per[False]), "delta_us": min(per[True]) - min(per[False]),
                  "what": "LM iteration with the back-substitution workgroups ACQUIRING the solve's release word (agent-scope fence: the HIP memory model's form, the default -- `value` is measured with it) against the relaxed form (agent-scope relaxed loads + in-order issue; MCBA_STRICT_SYNC=0); best of 3 alternating blocks of 200 iterations each"}
    if prof_exact[1]:
        prof[DOMINANT] = prof_exact
    else:   # (a launch variant without dispatch events: the bracketed figure)
        prof[DOMINANT] = prof_dom
    # ---- second figure of the metric: one materialised Jacobian evaluation (not part of the timed steps)
    prob.profile_enable(True)
    for _ in range(3):
        prob.jacobian_eval(lm.cur, robust_scaled=True)
    prob.profile_read()
    nj = 10
    for _ in range(nj):
        prob.jacobian_eval(lm.cur, robust_scaled=True)
    pj = prob.profile_read()["k_jacobian"]
    prob.profile_enable(False)
    ms_jac = pj[0] / pj[1]

    if rank == 0:
        kern = {k: (ms, n) for k, (ms, n) in prof.items() if n}
        dom = max(kern, key=lambda k: kern[k][0] / kern[k][1])
        dom_ms = kern[dom][0] / kern[dom][1]
        dom_bytes = algorithmic_bytes(dom, C, F, N)
        # HBM bytes from the PMC counters are NOT measured in this run (counter passes need rocprofv3): the figure printed is the
        # one of the committed profile named next to it, collected at the default shard size with the same kernels
        traffic = jtraffic = traffic_source = rocprof_stats = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file) and F == F_PER_GPU:
            with open(pmc_file) as fh:
                pmc = json.load(fh)
            rocprof_stats = pmc.get("_rocprofv3_kernel_stats")
            traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
            jtraffic = pmc.get("k_jacobian", {}).get("hbm_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json <- " + str(pmc.get("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier gpurun call")) + " (not measured in this run)"
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9
        hbm = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": dom_bytes}
        roofline = {"kernel": dom, "bound": "hbm", "n_samples": int(kern[dom][1]), **hbm, "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": dom_ms * 1e3}
        if dom == "k_gram":
            # the binding roof of this kernel is the FP64 vector pipe, not HBM (PMC traffic = 1.02 x algorithmic bytes).  `frac` =
            # REAL flops (2 FMA + MUL + ADD, as the SQ_INSTS_VALU_*_F64 counters count them) / the FP64 vector peak;
            # `frac_issue_slots` = every FP64 instruction priced as an FMA: how full the FP64 issue slots are, whatever the mix.
            # (The contract's "bound" enum has no FP64-vector entry; "valu_f64" names it -- on gfx950 the FP64 MFMA rate equals
            # the FP64 vector rate, so the dense-MFMA peak for this dtype is the same 78.6 TFLOP/s.)
            real, slots, mix = gram_work(C, F, N)
            tf, tfs = real / (dom_ms * 1e-3) / 1e12, slots / (dom_ms * 1e-3) / 1e12
            try:   # the FP64 vector rate this GPU sustains (independent v_fma_f64, one wavefront per SIMD on every CU), measured now
                live_ceiling = m.ops.fp64_issue_rate(local_rank)
            except Exception:  # noqa: BLE001
                live_ceiling = None
            roofline = {"kernel": dom, "bound": "valu_f64", "n_samples": int(kern[dom][1]), "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                        "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": dom_ms * 1e3, "avg_launch_us_by_events_raw": dom_events_us, "event_bracket_overhead_us": bracket_us,
                        "timed_region_samples": {"n": int(prof_timed[DOMINANT][1]), "avg_us_by_events_raw": (1e3 * prof_timed[DOMINANT][0] / prof_timed[DOMINANT][1]) if prof_timed[DOMINANT][1] else None,
                                                 "what": "every 8th launch inside the timed region (the brackets must not pace the stream)"},
                        "flops_per_launch": real,
                        "frac_issue_slots": tfs / FP64_VALU_PEAK_TFLOPS, "issue_slot_flops_per_launch": slots,
                        "measured_issue_ceiling": FP64_VALU_MEASURED_TFLOPS, "frac_issue_slots_of_measured_ceiling": tfs / FP64_VALU_MEASURED_TFLOPS,
                        "live_issue_ceiling": {"tflops": live_ceiling, "frac_issue_slots": (tfs / live_ceiling) if live_ceiling else None,
                                               "what": "independent v_fma_f64 with three distinct register pairs each, one wavefront per SIMD on every CU (mcba_fp64_issue_rate), measured in this run"},
                        "instructions_per_point_observation": mix, "hbm": hbm,
                        # rocprofv3's average of the same kernel in the committed profile of the same command (another process, another box): what
                        # the line's own figure should be read against -- the dispatch events of a plain run read 1-3 % above it
                        "rocprofv3": ({"avg_launch_us": rocprof_stats["kernels"][dom]["avg_us"], "calls": rocprof_stats["kernels"][dom]["calls"],
                                       "frac": real / (rocprof_stats["kernels"][dom]["avg_us"] * 1e-6) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                       "this_run_over_profile": dom_ms * 1e3 / rocprof_stats["kernels"][dom]["avg_us"], "source": rocprof_stats["source"] + " (not measured in this run)"}
                                      if rocprof_stats and dom in rocprof_stats.get("kernels", {}) else None),
                        "note": "frac = real FP64 flops (2 x FMA + MUL + ADD) / 78.6 TFLOP/s; frac_issue_slots counts every FP64 instruction as an FMA; avg_launch_us = mean of 50 launches right after the timed region, "
                                "timed by events attached to the kernel's dispatch (its own begin / end: the figure rocprofv3 reports); avg_launch_us_by_events_raw = the same launches with event records AROUND them"}
        tick_bytes = TICK_ALGORITHMIC_BYTES_10K * F / F_PER_GPU
        tick_ach = tick_bytes / (dt / args.steps) / 1e9
        tick_roofline = {"bound": "hbm", "achieved": tick_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tick_ach / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": tick_bytes,
                         "note": "SURVEY 8(d): 218 MB per fused LM iteration at 6 x 10 000 x 54 (scaled by frames) / ms_per_step of the timed region"}
        jb = algorithmic_bytes("k_jacobian", C, F, N)
        jach = jb / (ms_jac * 1e-3) / 1e9
        units = frames_total / F_PER_GPU  # 10k-frame problem units processed per step by all ranks together
        if scaling == "weak":
            value = args.steps * units / dt
            metric = "LM iterations/sec (6 cams x 10k frames x 54 pts per GPU shard; + ms/Jacobian-eval)"
            workload = f"bundle adjustment, {C} cameras x {F} frames/GPU x {N} points, intrinsics+distortion+extrinsics+poses free, soft_l1 (BASELINE configs[2])"
        else:
            value = args.steps / dt
            metric = f"LM iterations/sec of ONE {C} cams x {frames_total} frames x {N} pts problem, frames sharded over the GPUs (+ ms/Jacobian-eval of a shard)"
            workload = (f"bundle adjustment, {C} cameras x {frames_total} frames x {N} points sharded over {world} GPU(s) ({F} frames on rank 0), "
                        f"intrinsics+distortion+extrinsics+poses free, soft_l1 (BASELINE configs[3] when frames_total = 100000)")
        out = {
            "metric": metric,
            "value": value,
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "windows": len(window_dts),
            "value_windows": [(args.steps * units / w) if scaling == "weak" else (args.steps / w) for w in window_dts],
            "ms_per_step_windows": [1e3 * w / args.steps for w in window_dts],
            "value_is": "the median of the windows",
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "frames_total": frames_total, "frames_rank0": F,
                       "parallelism": f"frames sharded over {world} GPU(s), ONE all-reduce of the {12 * C}x{12 * C} reduced camera system (+ trial scalars) per iteration",
                       "collectives": type(comm).__name__ if comm is not None else "none",
                       "collective_backend": backend,
                       "ranks_seen_by_rccl": prob.comm_count() if comm is not None else 0,
                       "reduced_solver": "device (k_solve_cam), host two iterations ahead, no synchronisation per iteration" if lm.device_solve else "host (LAPACK), one synchronisation per iteration"},
            "value_in_10k_frame_units": args.steps * units / dt,
            "ms_per_jacobian_eval": ms_jac,
            "roofline": roofline,
            "tick_roofline": tick_roofline,
            "jacobian_eval": {"kernel": "k_jacobian", "ms": ms_jac, "roofline": {"bound": "hbm", "achieved": jach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": jach / HBM_PEAK_GBS,
                              "traffic": jtraffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": jb,
                              "frac_of_measured_write_ceiling": jach / HBM_MEASURED_WRITE_GBS}},
            "kernels_us": {k: round(1e3 * ms / n, 3) for k, (ms, n) in kern.items()},
            "kernel_calls": {k: n for k, (ms, n) in kern.items()},
            "strict_sync": strict,
            "prewarm": f"{args.prewarm} untimed iterations of a throw-away solve (clock ramp) before the {args.warmup} warm-up steps; the measured solve restarts from x0",
            "lm": {"cost_start": cost0, "cost_end": lm.cost, "accepted": lm.iteration, "steps_total": lm.steps, "lambda": lm.lam,
                   "passes_in_timed_region": ticks, "rebuild_only_passes_total": getattr(lm, "rebuilds", 0)},
        }
        if share_gpu:
            out["rehearsal"] = f"{world} ranks SHARE GPU 0 (fewer GPUs than ranks on this box): collectives are host-staged gloo all-reduces, not RCCL; `value` is not a scaling measurement"
        if world == 1 and not args.frames_total and F == F_PER_GPU and not args.no_other_configs:
            out["configs"] = other_configs(m, 1e3 * dt / args.steps)
        if world == 1 and not args.no_end_to_end and not args.frames_total:
            e2e = end_to_end(m, p)
            out["end_to_end_ms"] = e2e["ms"]
            out["end_to_end"] = e2e
            # ... and where users are: the one recording the reference documents (docs/source/calibration_tutorial.ipynb: 6 cameras x 2 130
            # frames x 35 points) and BASELINE configs[0]
            p_tut, p_c0 = m.synth.make_problem(6, 2130, rows=5, cols=7, seed=0), m.synth.make_problem(2, 50, rows=ROWS, cols=COLS, seed=0)
            out["end_to_end_other_shapes"] = {
                "reference tutorial 6 x 2130 x 35": end_to_end(m, p_tut, reps=7, full=False),
                "configs[0] 2 x 50 x 54": end_to_end(m, p_c0, reps=7, full=False)}
            # the call that produces bundle_adjust()'s inputs, and the pipeline calibrate() -> bundle_adjust()
            cal_main = calibrate_timing(m, p)
            e2e["calibrate_ms"], e2e["pipeline_ms"], e2e["calibrate"] = cal_main["calibrate_ms"], cal_main["pipeline_ms"], cal_main
            for key, pp in (("reference tutorial 6 x 2130 x 35", p_tut), ("configs[0] 2 x 50 x 54", p_c0)):
                ct = calibrate_timing(m, pp, reps=7)
                o = out["end_to_end_other_shapes"][key]
                o["calibrate_ms"], o["pipeline_ms"], o["calibrate"] = ct["calibrate_ms"], ct["pipeline_ms"], ct
            p_miss = m.synth.make_problem(C, F_PER_GPU, rows=ROWS, cols=COLS, seed=0, missing=0.3)
            e2e["calibrate_30pct_missing_detections"] = calibrate_timing(m, p_miss, reps=3)
            del p_miss
            # the off-default solver paths of the same call: bounds=, callable loss=
            e2e["off_default_paths"] = off_default_calls(m, p)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["reference_measured_in_survey_container"] = {"value": 0.0098, "unit": "it/s", "ms_per_jacobian_eval": 68679, "source": "BASELINE.md section 2 (the reference itself, 6x10kx54)"}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    prob.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
