"""`bundle_adjust` -- the reference's entry point (multicam_calibration/bundle_adjustment.py:195-327),
same signature, same 5-tuple, computed on the MI355X.

What is kept from the reference, line for line in behaviour:
  * frame pre-filter: frames complete in >= 2 cameras (:266), worst-camera mean reprojection error against
    5 x nan-median or `outlier_threshold` (:269-285), the printed "Excluding ..." line (:287-290, which
    reports the post-filter count), the global-RNG `np.random.choice` subsample (:293-296);
  * parameter layout of `result.x` (:128-157), `dist_coefs` padded to 5 with p1=p2=k3=0 on output (:187);
  * defaults verbose=2, x_scale='jac', ftol=1e-4, method='trf', loss='soft_l1', overridable through
    **opt_kwargs (:301-304); scipy's xtol = gtol = 1e-8 otherwise.
What differs: the optimiser is Levenberg-Marquardt with an analytic Jacobian and an
exact Schur solve instead of scipy's TRF/LSMR on a finite-difference Jacobian.  It minimises the same robust
cost, so it converges to the same minimiser (gauge aside); the iterate sequence is not reproduced.
"""
import warnings

import numpy as np
import scipy.sparse as sp

from . import ops, solver

# least_squares kwargs that only steer scipy's own iteration (no effect on the minimiser): accepted, ignored
_PATH_ONLY = ("jac", "tr_solver", "tr_options", "jac_sparsity", "diff_step", "x_scale", "method", "workers", "callback")


def serialize_params(all_extrinsics, all_intrinsics, calib_poses):
    """[C x (fx fy cx cy k1 k2 rx ry rz tx ty tz) | F x pose6]  (bundle_adjustment.py:128-157)."""
    cams = np.empty((len(all_extrinsics), 12))
    for c, (ext, (K, dist)) in enumerate(zip(all_extrinsics, all_intrinsics)):
        K = np.asarray(K, dtype=float)
        cams[c, :4] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        cams[c, 4:6] = np.asarray(dist, dtype=float)[:2]
        cams[c, 6:] = ext
    return np.concatenate([cams.ravel(), np.asarray(calib_poses, dtype=float).ravel()])


def deserialize_params(x, n_cameras):
    """Inverse of serialize_params (bundle_adjustment.py:160-192)."""
    cams = np.asarray(x[: 12 * n_cameras]).reshape(n_cameras, 12)
    intr = []
    for c in range(n_cameras):
        K = np.eye(3)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = cams[c, :4]
        intr.append((K, np.pad(cams[c, 4:6], (0, 3))))
    return cams[:, 6:].copy(), intr, np.asarray(x[12 * n_cameras :]).reshape(-1, 6).copy()


def jacobian_structure(all_calib_uvs):
    """CSR (indices, indptr, shape) of the reference's sparsity pattern (bundle_adjustment.py:101-125)."""
    C, F, N, _ = all_calib_uvs.shape
    mask = ~np.isnan(all_calib_uvs)
    cam = np.broadcast_to(np.arange(C, dtype=np.int32)[:, None, None, None], mask.shape)[mask]
    frm = np.broadcast_to(np.arange(F, dtype=np.int32)[None, :, None, None], mask.shape)[mask]
    m = cam.size
    idx = np.empty((m, 18), dtype=np.int32 if 12 * C + 6 * F < 2**31 else np.int64)
    idx[:, :12] = cam[:, None] * 12 + np.arange(12)
    idx[:, 12:] = 12 * C + frm[:, None] * 6 + np.arange(6)
    return idx.ravel(), np.arange(m + 1, dtype=np.int64) * 18, (m, 12 * C + 6 * F), mask


def select_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames, outlier_threshold, device=0, backend=None, keep_problem=False, **problem_kw):
    """The reference's pre-filter (bundle_adjustment.py:265-296) on the GPU: ALL frames are uploaded once, the reprojection
    errors, their per-(camera, frame) nan-means, the completeness counts (k_frame_err) and the exact nan-median (radix
    select) are computed there; the host sees 2 x (C,F) doubles.  Returns use_frames, or (use_frames, problem) with
    keep_problem=True -- the handle that still holds every frame, for `Problem.subset` (no second upload)."""
    all_calib_uvs = np.asarray(all_calib_uvs, dtype=np.float64)
    C, F_all, N = all_calib_uvs.shape[:3]
    Problem = backend or ops.Problem
    prob = None
    if F_all and hasattr(Problem, "frame_errors"):
        prob = Problem(all_calib_uvs, calib_objpoints, device=device, **problem_kw)
        prob.set_params(0, serialize_params(all_extrinsics, all_intrinsics, np.asarray(calib_poses, dtype=np.float64)))
        mean_cf, full_cf = prob.frame_errors(0)
        use_frames = np.nonzero((full_cf == N).sum(0) > 1)[0]                     # complete in at least two cameras (:266)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            worst_mean_err = np.nanmax(mean_cf[:, use_frames], axis=0) if use_frames.size else np.empty(0)   # (:279)
        if outlier_threshold is None:
            mask = np.zeros(F_all, dtype=np.uint8)
            mask[use_frames] = 1
            outlier_threshold = 5 * prob.error_median(mask)[0]                    # 5 * np.nanmedian(err)  (:281-282)
    else:  # CPU test double (tests/fake_problem.py) or no frames at all: the same arithmetic in numpy
        use_frames = np.nonzero((~np.isnan(all_calib_uvs).any((-1, -2))).sum(0) > 1)[0]
        sub = all_calib_uvs[:, use_frames]
        if use_frames.size:
            fake = Problem(sub, calib_objpoints, device=device)
            fake.set_params(0, serialize_params(all_extrinsics, all_intrinsics, np.asarray(calib_poses)[use_frames]))
            res = fake.residuals(0)
            fake.close()
            res[np.isnan(sub)] = np.nan
            err = np.sqrt((res**2).sum(-1))  # NaN wherever either coordinate is missing, like norm(obs - pred)
        else:
            err = np.empty(sub.shape[:-1])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            worst_mean_err = np.nanmax(np.nanmean(err, axis=-1), axis=0) if use_frames.size else np.empty(0)
            if outlier_threshold is None:
                outlier_threshold = 5 * np.nanmedian(err)
    exclude = np.nan_to_num(worst_mean_err) > outlier_threshold
    use_frames = use_frames[~exclude]
    print(f"Excluding {int(exclude.sum())} out of {len(use_frames)} frames based on an outlier threshold of {outlier_threshold}")
    if not (n_frames is None or n_frames > len(use_frames)):
        use_frames = np.random.choice(use_frames, n_frames, replace=False)
    if keep_problem:  # + whether every selected detection is complete (then no NaN mask is needed for result.fun / result.jac)
        complete = prob is not None and bool(np.all(full_cf[:, use_frames] == N))
        return use_frames, prob, complete
    if prob is not None:
        prob.close()
    return use_frames


def bundle_adjust(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames=10000, outlier_threshold=None, **opt_kwargs):
    """Bundle adjustment for camera parameters and calibration-object poses on the MI355X.

    Parameters and return value are those of the reference `multicam_calibration.bundle_adjust`.
    Extra keyword arguments (defaults preserve reference behaviour):
      device=0             HIP device ordinal (distributed: defaults to LOCAL_RANK)
      fix_intrinsics=False hold fx fy cx cy k1 k2 of every camera (BASELINE config 2); extrinsics + poses only
      return_jac=True      attach the robust-rescaled CSR Jacobian as `result.jac` (18 nnz/row; 1.4 GB at 6x10k x54)
      distributed=False    one process per GPU under torch.distributed (an initialised process group, backend nccl =
                           RCCL): the frames chosen by rank 0 are sharded contiguously over the ranks, every LM iteration
                           all-reduces the reduced camera system, and every rank returns the full 5-tuple (poses
                           all-gathered).  `result.fun` / `result.jac` then cover the calling rank's shard only.
    """
    distributed = opt_kwargs.pop("distributed", False)
    backend = opt_kwargs.pop("_backend", None)  # test hook: a drop-in for ops.Problem (tests/fake_problem.py)
    rank, world, dist = 0, 1, None
    if distributed:
        import os
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("distributed=True needs an initialised torch.distributed process group (launch with torchrun)")
        rank, world = dist.get_rank(), dist.get_world_size()
        opt_kwargs.setdefault("device", int(os.environ.get("LOCAL_RANK", rank)))
        if backend is None:
            import torch

            torch.cuda.set_device(opt_kwargs["device"])  # object collectives of the nccl backend use the current device
    device = opt_kwargs.pop("device", 0)
    fix_intrinsics = opt_kwargs.pop("fix_intrinsics", False)
    return_jac = opt_kwargs.pop("return_jac", True)
    lm_kwargs = {k: opt_kwargs.pop(k) for k in ("lam0", "reduced_solver") if k in opt_kwargs}

    all_calib_uvs = np.asarray(all_calib_uvs, dtype=np.float64)
    calib_objpoints = np.asarray(calib_objpoints, dtype=np.float64)
    calib_poses = np.asarray(calib_poses, dtype=np.float64)
    n_cameras = all_calib_uvs.shape[0]

    prob_all, all_seen = None, False
    if not distributed:
        use_frames, prob_all, all_seen = select_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames, outlier_threshold, device, backend, keep_problem=True)
    else:
        # rank 0 owns the reference's frame selection (its printed line and its use of the global numpy RNG)
        box = [select_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames, outlier_threshold, device, backend) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        use_frames = box[0]

    kw = dict(verbose=2, x_scale="jac", ftol=1e-4, method="trf", loss="soft_l1")
    kw.update(opt_kwargs)
    if "bounds" in kw:
        raise NotImplementedError("bounds are not supported by the GPU solver")
    if callable(kw.get("loss")):
        raise NotImplementedError("callable losses are not supported by the GPU solver")
    if not (isinstance(kw["x_scale"], str) and kw["x_scale"] == "jac"):
        warnings.warn("x_scale is ignored: the GPU solver always uses Jacobian (Marquardt) scaling", stacklevel=2)
    unknown = set(kw) - set(_PATH_ONLY) - {"verbose", "ftol", "xtol", "gtol", "max_nfev", "loss", "f_scale"}
    if unknown:
        raise TypeError(f"unsupported least_squares keyword(s) for the GPU solver: {sorted(unknown)}")

    all_use = use_frames
    if distributed:
        if use_frames.size < world:
            raise ValueError(f"{use_frames.size} usable frames cannot be sharded over {world} ranks")
        use_frames = np.array_split(all_use, world)[rank]  # contiguous shard of the selection, in selection order
    x0 = serialize_params(all_extrinsics, all_intrinsics, calib_poses[use_frames])
    if use_frames.size == 0:
        if prob_all is not None:
            prob_all.close()
        # nothing to fit: scipy's least_squares on an empty residual vector returns x0 with status 1 (gtol) after one
        # evaluation (what the reference then returns: bundle_adjustment.py:307-327)
        from scipy.optimize import OptimizeResult

        result = OptimizeResult(x=x0, cost=0.0, fun=np.empty(0), jac=sp.csr_matrix((0, x0.size)), grad=np.zeros(x0.size), optimality=0.0,
                                active_mask=np.zeros(x0.size), nfev=1, njev=1, status=1, message=solver.TERMINATION_MESSAGES[1], success=True)
        if kw["verbose"] >= 1:
            print(result.message)
            print("Function evaluations 1, initial cost 0.0000e+00, final cost 0.0000e+00, first-order optimality 0.00e+00.")
        ext, intr, poses = deserialize_params(x0, n_cameras)
        return ext, intr, poses, use_frames, result
    pkw = {}
    if distributed and backend is None:
        import torch

        # torch orders its collectives against torch's CURRENT stream: the library must launch on that same stream
        # (the torch.distributed fallback of solver.make_comm all-reduces the library's reduce buffer in place)
        pkw["stream"] = torch.cuda.current_stream(device).cuda_stream
    if prob_all is not None:   # the frames are already on the GPU (pre-filter): gather the selection there, no second upload
        prob = prob_all.subset(use_frames, loss=kw["loss"], f_scale=kw.get("f_scale", 1.0))
        prob_all.close()
        uvs = None             # host copy of the selection: only materialised if result.jac / result.fun need the NaN mask
    else:
        uvs = np.ascontiguousarray(all_calib_uvs[:, use_frames])
        prob = (backend or ops.Problem)(uvs, calib_objpoints, device=device, loss=kw["loss"], f_scale=kw.get("f_scale", 1.0), **pkw)
    comm = None
    if distributed:
        import torch

        comm = solver.make_comm(prob, torch.device(f"cuda:{device}") if backend is None else "cpu")
        return_jac = False if backend is not None else return_jac
    free = None
    if fix_intrinsics:
        free = np.tile(np.r_[np.zeros(6, bool), np.ones(6, bool)], n_cameras)
    # an explicit None disables that test, as scipy's check_tolerance does (least_squares.py: None -> 0); the device loop
    # treats a zero tolerance as "never satisfied".  A missing key keeps the reference's / scipy's defaults.
    tol = lambda name, default: 0.0 if kw.get(name, default) is None else float(kw.get(name, default))
    max_nfev = kw.get("max_nfev")
    if max_nfev is None:
        max_nfev = 100 * (12 * n_cameras + 6 * len(all_use))  # trf.py:437-438 on the GLOBAL vector: identical on every rank
    result = solver.lm_solve(prob, x0, ftol=tol("ftol", 1e-4), xtol=tol("xtol", 1e-8), gtol=tol("gtol", 1e-8),
                             max_nfev=max_nfev, verbose=kw["verbose"] if rank == 0 else 0, free_cam_mask=free, comm=comm, **lm_kwargs)

    # ---- OptimizeResult fields the reference's callers can rely on (trf.py:557-560)
    slot = result.lm["slot"]
    if uvs is None and not (all_seen and not return_jac):
        uvs = all_calib_uvs[:, use_frames]
    if backend is None and return_jac:
        idx, indptr, shape, mask = jacobian_structure(uvs)  # CSR indices: 0.25 s of numpy at 6 x 10k x 54 -- only when asked for
    elif uvs is not None:
        mask = ~np.isnan(uvs)
    if uvs is None:  # every selected detection is complete (the pre-filter counted them on the GPU): no NaN mask to apply
        result.fun = prob.residuals(slot).ravel()
    elif backend is not None:  # test double: no materialised Jacobian kernel
        result.fun = prob.residuals(slot)[mask]
    elif return_jac:
        prob.jacobian_eval(slot, robust_scaled=kw["loss"] != "linear")
        jac, res = prob.jacobian_download()
        result.jac = sp.csr_matrix((jac[mask].ravel(), idx, indptr), shape=shape)
        result.fun = res[mask]
        del jac
    else:
        result.fun = prob.residuals(slot)[mask]
    red = prob.get_reduced()
    grad = np.concatenate([red["gc"], prob.frame_gradient().ravel()])
    if free is not None:
        grad[: 12 * n_cameras][~free] = 0.0
    if distributed:
        # assemble the global vectors: cameras are identical on every rank, poses / frame gradients are gathered
        parts = [None] * world
        dist.all_gather_object(parts, (result.x[12 * n_cameras:], grad[12 * n_cameras:]))
        result.x = np.concatenate([result.x[: 12 * n_cameras]] + [p[0] for p in parts])
        grad = np.concatenate([grad[: 12 * n_cameras]] + [p[1] for p in parts])
        result.active_mask = np.zeros_like(result.x)
        result.optimality = float(np.abs(grad).max())
    result.grad = grad
    prob.close()

    adjusted_extrinsics, adjusted_intrinsics, adjusted_calib_poses = deserialize_params(result.x, n_cameras)
    return adjusted_extrinsics, adjusted_intrinsics, adjusted_calib_poses, all_use, result


# BASELINE.json's north star calls the entry point `bundle_adjustment()`; the reference function is `bundle_adjust`.
bundle_adjustment = bundle_adjust
