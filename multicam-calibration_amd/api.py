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
import os
import weakref

import numpy as np
import scipy.sparse as sp
from scipy.optimize import OptimizeResult

from . import ops, solver

# least_squares kwargs that only steer scipy's own iteration (no effect on the minimiser): accepted, ignored
_PATH_ONLY = ("jac", "tr_solver", "tr_options", "jac_sparsity", "diff_step", "method", "workers", "callback")


def serialize_params(all_extrinsics, all_intrinsics, calib_poses):
    """[C x (fx fy cx cy k1 k2 rx ry rz tx ty tz) | F x pose6]  (bundle_adjustment.py:128-157)."""
    C = len(all_extrinsics)
    poses = np.asarray(calib_poses, dtype=np.float64)
    x = np.empty(12 * C + poses.size)
    cams = x[: 12 * C].reshape(C, 12)
    for c, (ext, (K, dist)) in enumerate(zip(all_extrinsics, all_intrinsics)):
        K = np.asarray(K, dtype=float)
        cams[c, 0], cams[c, 1], cams[c, 2], cams[c, 3] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        cams[c, 4:6] = np.asarray(dist, dtype=float)[:2]
        cams[c, 6:] = ext
    x[12 * C:] = poses.ravel()
    return x


def deserialize_params(x, n_cameras):
    """Inverse of serialize_params (bundle_adjustment.py:160-192)."""
    cams = np.asarray(x[: 12 * n_cameras]).reshape(n_cameras, 12)
    Ks = np.zeros((n_cameras, 3, 3))
    Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2], Ks[:, 2, 2] = cams[:, 0], cams[:, 1], cams[:, 2], cams[:, 3], 1.0
    dists = np.zeros((n_cameras, 5))
    dists[:, :2] = cams[:, 4:6]
    intr = [(Ks[c].copy(), dists[c].copy()) for c in range(n_cameras)]   # (arrays of their own, as the reference returns them)
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


class _Lazy:
    """Placeholder for an OptimizeResult field that is produced on first access."""

    def __init__(self, thunk):
        self.thunk = thunk


class _JacobianSource:
    """What a lazily attached `result.jac` is produced from: the GPU handle of the solve, trimmed to what a Jacobian needs (observations,
    board, parameters: ops.Problem.trim -- 0.11 GB at 6 x 10 000 x 54) while it lasts.  The handles that pending results hold are
    counted by what they REALLY hold (ops.Problem.device_bytes): beyond MCBA_JAC_HOLD_MB (default 1024 MiB; 0 = hold nothing) the oldest
    are closed at once -- a sweep that keeps a list of results does not pin HBM per result -- and a source that has to let
    its handle go first copies the observations back FROM THE GPU: the values the solve saw.  A later read re-creates the handle from
    that copy, never from the caller's array, which may have been changed in place since the call."""

    _refs = []   # weak references to the sources that still own a handle, oldest first (a dropped result frees its handle by itself)

    class _Live:
        def __get__(self, obj, cls):
            alive = [r() for r in cls._refs]
            cls._refs = [r for r, a in zip(cls._refs, alive) if a is not None and a.prob is not None]
            return [a for a in alive if a is not None and a.prob is not None]

    live = _Live()

    def __init__(self, prob, objpoints, x, slot, loss, f_scale, device, robust, need_mask, shape4):
        self.prob, self.obj, self.x, self.slot = prob, objpoints, x, slot
        self.loss, self.f_scale, self.device, self.robust, self.need_mask, self.shape4 = loss, f_scale, device, robust, need_mask, shape4
        self.host_uvs = None
        if hasattr(prob, "trim"):
            prob.trim()
        self.bytes = prob.device_bytes() if hasattr(prob, "device_bytes") else 3 * 8 * int(np.prod(shape4)) + 16 * x.size
        _JacobianSource._refs.append(weakref.ref(self))
        _JacobianSource.trim()

    @classmethod
    def trim(cls):
        cap = int(os.environ.get("MCBA_JAC_HOLD_MB", "1024")) << 20
        live = cls.live
        while live and sum(s.bytes for s in live) > cap:
            live.pop(0).release(keep_values=True)

    def release(self, keep_values=False):
        if self.prob is not None:
            if keep_values and self.host_uvs is None and hasattr(self.prob, "download_observations"):
                self.host_uvs = self.prob.download_observations()   # (from the device copy: what the solve saw)
            self.prob.close()
            self.prob = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001
            pass

    def mask(self, prob):
        """Which scalars were observed: from the GPU's copy of the observations (immutable since the call), when it is needed."""
        if not self.need_mask:
            return None
        if hasattr(prob, "seen_bits"):
            return np.unpackbits(prob.seen_bits(), count=int(np.prod(self.shape4))).astype(bool).reshape(self.shape4)
        return ~np.isnan(prob.uvs)   # (the CPU test double keeps its observations as an array)

    def csr(self):
        try:
            prob, slot = self.prob, self.slot
            if prob is None:   # the handle was released under pressure: a new one from the values it held
                if self.host_uvs is None:
                    raise RuntimeError("result.jac: the GPU handle of this result was released and no copy of its observations was kept")
                prob = self.prob = ops.Problem(self.host_uvs, self.obj, device=self.device, loss=self.loss, f_scale=self.f_scale)
                prob.set_params(0, self.x)
                slot = 0
            m = self.mask(prob)
            uvs = np.zeros(self.shape4) if m is None else np.where(m, 0.0, np.nan)   # (jacobian_structure only looks at which scalars are NaN)
            idx, indptr, shape, mask = jacobian_structure(uvs)  # CSR indices: 0.25 s of numpy at 6 x 10k x 54 -- only when asked for
            own_rho = callable(self.loss)   # least_squares' callable loss: unscaled rows from the GPU, scipy's row scale from the caller's function (common.py:720-731)
            prob.jacobian_eval(slot, robust_scaled=self.robust and not own_rho)
            data, _ = prob.jacobian_download(want_res=False)
            data = data[mask]
            if own_rho:
                _, z, rho = prob.loss_values(slot)
                data *= np.sqrt(np.maximum(rho[1] + 2.0 * rho[2] * z, np.finfo(float).eps))[:, None]
            return sp.csr_matrix((data.ravel(), idx, indptr), shape=shape)
        finally:
            self.release()
            self.host_uvs = None


class LazyOptimizeResult(OptimizeResult):
    """scipy's OptimizeResult (a dict with attribute access) whose expensive fields are materialised when first read.
    `result.fun` (the residual vector at the solution, trf.py:557-560) stays on the GPU until then: 52 MB of device-to-host copy at
    6 x 10 000 x 54 that most callers never look at.  Every way of reading the field (attribute, item, get, items, values, repr,
    copy, pickling) goes through `_resolve`, after which the object is an ordinary OptimizeResult."""

    def _resolve(self, key=None):
        for k in ([key] if key is not None else list(dict.keys(self))):
            v = dict.get(self, k)
            if isinstance(v, _Lazy):
                dict.__setitem__(self, k, v.thunk())

    def __getitem__(self, key):
        self._resolve(key)
        return dict.__getitem__(self, key)

    def __iter__(self):
        # (defined so that dict(result), {**result} and OptimizeResult(result) leave CPython's fast path for dict subclasses -- it
        #  copies the raw slots, placeholders included -- and go through keys() + __getitem__, which resolves each field)
        return dict.__iter__(self)

    def get(self, key, default=None):
        self._resolve(key)
        return dict.get(self, key, default)

    def items(self):
        self._resolve()
        return dict.items(self)

    def values(self):
        self._resolve()
        return dict.values(self)

    def copy(self):
        self._resolve()
        return OptimizeResult(dict.copy(self))

    def __repr__(self):
        self._resolve()
        return OptimizeResult.__repr__(self)

    def __reduce__(self):
        self._resolve()
        return (OptimizeResult, (dict(self),))


def _split_bounds(n, world):
    """Contiguous slices of range(n), one per rank (np.array_split's sizes)."""
    sizes = [n // world + (1 if r < n % world else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def _median_from_histograms(hist):
    """np.nanmedian by radix select: hist(prefix, pass) -> 256 global counts of the next byte among the values with that prefix
    (the per-point errors are non-negative doubles, so the order of their bit patterns is their numeric order); the two middle order
    statistics are averaged like np.median.  Integer arithmetic only: exact whatever the sharding."""
    first = hist(0, 0)
    total = int(first.sum())
    if total == 0:
        return float("nan")
    vals = []
    for rank in ((total - 1) // 2, total // 2):
        prefix, r = 0, rank
        for p in range(8):
            h = first if p == 0 else hist(prefix, p)
            below = np.concatenate([[0], np.cumsum(h.astype(np.int64))])
            digit = int(np.searchsorted(below, r, side="right")) - 1
            r -= int(below[digit])
            prefix = (prefix << 8) | digit
        vals.append(np.array([prefix], dtype=np.uint64).view(np.float64)[0])
        if rank == total // 2:
            break
    return float(0.5 * (vals[0] + vals[-1]))


def select_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames, outlier_threshold, device=0, keep_problem=False, group=None, subset_loss=None, **problem_kw):
    """The reference's pre-filter (bundle_adjustment.py:265-296) on the GPU: the frames are uploaded once, the reprojection
    errors, their per-(camera, frame) nan-means, the completeness counts (k_frame_err) and the exact nan-median (radix
    select) are computed there; the host sees 2 x (C,F) doubles.  Returns use_frames, or (use_frames, problem, complete, lo)
    with keep_problem=True -- the handle that still holds the frames, for `Problem.subset` (no second upload).

    group: a torch.distributed process group -> the FRAME-SHARDED pre-filter: every rank uploads and scores only its contiguous
    slice [lo, hi) of the frames (the handle returned holds that slice), the (C,F) statistics are all-gathered, the median's
    radix-select histograms are all-reduced (8 passes x 256 integer bins per order statistic), and rank 0 alone prints the
    reference's line and draws the subsample from the global numpy RNG (:293-296); the selection is broadcast."""
    C, F_all, N = all_calib_uvs.shape[:3]
    if group is None and F_all > 0 and hasattr(ops.Problem, "prefilter") and os.environ.get("MCBA_PREFILTER_FUSED", "1") != "0":
        # one GPU: upload + scoring + the whole selection in ONE C-ABI crossing and one host synchronisation (mcba_prefilter); what is
        # left for the host is what must happen there -- the printed line and the draw from the GLOBAL numpy RNG (:287-296)
        prob = ops.Problem(all_calib_uvs, calib_objpoints, device=device, upload=False, **problem_kw)
        try:
            # (a NaN threshold compares False with everything -- nothing is excluded --: +inf does the same and leaves NaN free to mean "5 x median" at the ABI)
            thr_in = None if outlier_threshold is None else (float("inf") if np.isnan(float(outlier_threshold)) else float(outlier_threshold))
            sub = None
            if keep_problem and subset_loss is not None and hasattr(prob, "prefilter_subset") and os.environ.get("MCBA_PREFILTER_SUBSET", "1") != "0":   # (0: gather in a crossing of its own, for A/B)
                # bundle_adjust: the frames kept are gathered in the same crossing unless the reference's random draw stands in between (:292-296)
                status, thr, info, sub = prob.prefilter_subset(serialize_params(all_extrinsics, all_intrinsics, calib_poses), thr_in, n_frames, *subset_loss)
            else:
                status, thr, info = prob.prefilter(serialize_params(all_extrinsics, all_intrinsics, calib_poses), thr_in)
        except BaseException:
            prob.close()
            raise
        use_frames = np.flatnonzero((status & 3) == 1)   # used (:266) and not excluded (:285); the counts come with the call (info 4..6)
        shown = thr if outlier_threshold is None else outlier_threshold
        print(f"Excluding {int(info[5])} out of {len(use_frames)} frames based on an outlier threshold of {shown}")
        all_seen = info[6] == 0.0   # no kept frame is incomplete in any camera
        if not (n_frames is None or n_frames > len(use_frames)):
            use_frames = np.random.choice(use_frames, n_frames, replace=False)
            all_seen = all_seen or bool((status[use_frames] & 4).all())
        if keep_problem:
            if subset_loss is not None:
                return use_frames, prob, bool(all_seen), 0, sub
            return use_frames, prob, bool(all_seen), 0
        prob.close()
        return use_frames
    dist, rank, world = None, 0, 1
    if group is not None:
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        coll_dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    bounds = _split_bounds(F_all, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    prob = None
    mean_cf = full_cf = np.empty((C, 0))
    try:
        if hi > lo:
            prob = ops.Problem(all_calib_uvs[:, lo:hi], calib_objpoints, device=device, **problem_kw)
            prob.set_params(0, serialize_params(all_extrinsics, all_intrinsics, calib_poses[lo:hi]))
            mean_cf, full_cf = prob.frame_errors(0)
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, (mean_cf, full_cf), group=group)
            mean_cf = np.concatenate([p[0] for p in parts], axis=1)
            full_cf = np.concatenate([p[1] for p in parts], axis=1)
        complete_cf = full_cf == N
        use_frames = np.nonzero(complete_cf.sum(0) > 1)[0]                        # complete in at least two cameras (:266)
        every = use_frames.size == F_all                                          # (no fancy-index copies in the common case)
        # np.nanmax over the cameras (:279) = fmax-reduction: NaN only where every camera is NaN, and no warning for it
        worst_mean_err = np.fmax.reduce(mean_cf if every else mean_cf[:, use_frames], axis=0) if use_frames.size else np.empty(0)
        if outlier_threshold is None:                                             # 5 * np.nanmedian(err)  (:281-282)
            mask = np.zeros(hi - lo, dtype=np.uint8)
            mask[use_frames[(use_frames >= lo) & (use_frames < hi)] - lo] = 1
            if dist is None:
                outlier_threshold = 5 * prob.error_median(mask)[0] if prob is not None else float("nan")
            else:
                state = {"first": True}

                def hist(prefix, p):
                    h = np.zeros(256, dtype=np.uint64)
                    if prob is not None:
                        h = prob.error_histogram(mask if state["first"] else None, prefix, p)
                        state["first"] = False
                    t = torch.from_numpy(h.astype(np.int64)).to(coll_dev)
                    dist.all_reduce(t, group=group)
                    return t.cpu().numpy()

                outlier_threshold = 5 * _median_from_histograms(hist)
    except BaseException:
        if prob is not None:
            prob.close()
        raise
    exclude = np.nan_to_num(worst_mean_err) > outlier_threshold
    use_frames = use_frames[~exclude]
    if rank == 0:
        print(f"Excluding {int(exclude.sum())} out of {len(use_frames)} frames based on an outlier threshold of {outlier_threshold}")
        if not (n_frames is None or n_frames > len(use_frames)):
            use_frames = np.random.choice(use_frames, n_frames, replace=False)
    if dist is not None:
        box = [use_frames]
        dist.broadcast_object_list(box, src=0, group=group)
        use_frames = box[0]
    if keep_problem:  # + whether every selected detection is complete (then no NaN mask is needed for result.fun / result.jac)
        complete = bool(complete_cf.all()) or bool(complete_cf[:, use_frames].all())
        return use_frames, prob, complete, lo
    if prob is not None:
        prob.close()
    return use_frames


def _check_x_scale(x_scale, n_total):
    """least_squares.py:  `x_scale` must be 'jac' or array_like with positive numbers; a scalar is broadcast."""
    if isinstance(x_scale, str) and x_scale == "jac":
        return None
    try:
        xs = np.asarray(x_scale, dtype=np.float64)
        valid = bool(np.all(np.isfinite(xs)) and np.all(xs > 0))
    except (ValueError, TypeError):
        valid = False
    if not valid:
        raise ValueError("`x_scale` must be 'jac' or array_like with positive numbers.")
    if xs.ndim == 0:
        xs = np.resize(xs, n_total)
    if xs.shape != (n_total,):
        raise ValueError("Inconsistent shapes between `x_scale` and `x0`.")
    return xs


def bundle_adjust(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames=10000, outlier_threshold=None, **opt_kwargs):
    """Bundle adjustment for camera parameters and calibration-object poses on the MI355X.

    Parameters and return value are those of the reference `multicam_calibration.bundle_adjust`.
    Extra keyword arguments (defaults preserve reference behaviour):
      device=0             HIP device ordinal (distributed: defaults to LOCAL_RANK)
      fix_intrinsics=False hold fx fy cx cy k1 k2 of every camera (BASELINE config 2); extrinsics + poses only
      lam0=1e-3, dec_floor=0.1   damping schedule of the LM loop (solver.py); dec_floor=1/3 is Nielsen's classical rule
      curvature="auto"           curvature model of the linearisations (solver.py: CURV_SWITCH): "auto" = the IRLS weight rho' until an accepted step
                                 gains less than 1 % of the cost, then Triggs' second-order term; "irls" / "triggs" = one of them throughout
      return_jac=True      attach the robust-rescaled CSR Jacobian as `result.jac` (18 nnz/row; 1.4 GB at 6x10k x54) -- lazily: it is
                           produced when the field is first read; until then the result keeps the GPU handle alive
                           (return_jac=False releases it at once)
      distributed=False    one process per GPU under torch.distributed (an initialised process group, backend nccl =
                           RCCL): every rank uploads and pre-filters its contiguous slice of ALL frames, solves the selected
                           frames of that slice, every LM iteration all-reduces the reduced camera system, and every rank
                           returns the full 5-tuple (poses gathered in selection order).  `result.fun` / `result.jac` then
                           cover the calling rank's frames only (`result.lm["frame_positions"]` = their places in use_frames).
    `result.fun` and `result.jac` are produced when they are first read (LazyOptimizeResult).  `result.lm` carries the solver's own
    record: iterations, damping history, which collective backend and reduced solver ran.
    """
    distributed = opt_kwargs.pop("distributed", False)
    rank, world, dist, group = 0, 1, None, None
    if distributed:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("distributed=True needs an initialised torch.distributed process group (launch with torchrun)")
        group = dist.group.WORLD
        rank, world = dist.get_rank(), dist.get_world_size()
        opt_kwargs.setdefault("device", int(os.environ.get("LOCAL_RANK", rank)))
        if dist.get_backend() == "nccl":
            torch.cuda.set_device(opt_kwargs["device"])  # object collectives of the nccl backend use the current device
    device = opt_kwargs.pop("device", 0)
    fix_intrinsics = opt_kwargs.pop("fix_intrinsics", False)
    return_jac = opt_kwargs.pop("return_jac", True)
    lm_kwargs = {k: opt_kwargs.pop(k) for k in ("lam0", "dec_floor", "reduced_solver", "curvature") if k in opt_kwargs}

    kw = dict(verbose=2, x_scale="jac", ftol=1e-4, method="trf", loss="soft_l1")
    kw.update(opt_kwargs)
    box = kw.pop("bounds", None)
    unknown = set(kw) - set(_PATH_ONLY) - {"verbose", "ftol", "xtol", "gtol", "max_nfev", "loss", "f_scale", "x_scale"}
    if unknown:
        raise TypeError(f"unsupported least_squares keyword(s) for the GPU solver: {sorted(unknown)}")

    all_calib_uvs = np.asarray(all_calib_uvs, dtype=np.float64)
    calib_objpoints = np.asarray(calib_objpoints, dtype=np.float64)
    calib_poses = np.asarray(calib_poses, dtype=np.float64)
    n_cameras = all_calib_uvs.shape[0]
    # The kernels' camera model is (fx fy cx cy k1 k2) -- all the reference ever optimises (serialize_params drops the rest:
    # bundle_adjustment.py:149-155) -- but its PRE-FILTER projects with the full matrix (geometry.py:323: K @ p): a skew K[0,1] or a
    # non-trivial third row would select other frames there than here.  Refused rather than silently ignored.
    for c, (K, _) in enumerate(all_intrinsics):
        K = np.asarray(K, dtype=np.float64)
        if K.shape != (3, 3) or K[0, 1] != 0.0 or K[1, 0] != 0.0 or K[2, 0] != 0.0 or K[2, 1] != 0.0 or K[2, 2] != 1.0:
            raise ValueError(f"camera {c}: the camera matrix must be [[fx, 0, cx], [0, fy, cy], [0, 0, 1]] (skew / a general third row are not supported by the GPU solver; "
                             "the reference's pre-filter would honour them, geometry.py:323)")

    pkw = {}
    if distributed:
        import torch

        if torch.cuda.is_available():
            # torch orders its collectives against torch's CURRENT stream: the library must launch on that same stream
            # (the torch.distributed fallback of solver.make_comm all-reduces the library's reduce buffer in place)
            pkw["stream"] = torch.cuda.current_stream(device).cuda_stream
    sel = select_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames, outlier_threshold, device, keep_problem=True, group=group,
                        subset_loss=None if distributed else (kw["loss"], kw.get("f_scale", 1.0)), **pkw)
    all_use, prob_all, all_seen, lo = sel[:4]
    pre_sub = sel[4] if len(sel) > 4 else None   # (one GPU, no random draw: the kept frames were gathered in the pre-filter's own crossing)
    prob = None
    try:
        # ---- which of the selected frames this process solves: all of them, or (frame-sharded) those of its own slice -- they are
        # already on this GPU.  If a rank's slice holds none of the selection the shards are cut from the selection instead and
        # uploaded from the host array.
        positions = np.arange(all_use.size)
        local = True
        if distributed:
            if all_use.size < world:
                raise ValueError(f"{all_use.size} usable frames cannot be sharded over {world} ranks")
            bounds = _split_bounds(all_calib_uvs.shape[1], world)
            owner = np.searchsorted(bounds, all_use, side="right") - 1
            local = bool(np.bincount(owner, minlength=world).min() > 0)
            positions = np.nonzero(owner == rank)[0] if local else np.array_split(np.arange(all_use.size), world)[rank]
        use_frames = all_use[positions]
        nx_local = 12 * n_cameras + 6 * use_frames.size

        class _X0:   # the start vector of this process's frames, built when (if) somebody needs its VALUES: gathering 10 000 poses costs 0.15 ms on the
            v = None   # host, and the default path never looks at them (mcba_create_subset gathers the start point on the GPU)

            @classmethod
            def get(cls):
                if cls.v is None:
                    cls.v = serialize_params(all_extrinsics, all_intrinsics, calib_poses[use_frames])
                return cls.v

        x_scale = _check_x_scale(kw["x_scale"], 12 * n_cameras + 6 * all_use.size)
        if x_scale is not None:  # this shard's part: the camera block and its own frames' blocks
            x_scale = np.concatenate([x_scale[: 12 * n_cameras], x_scale[12 * n_cameras:].reshape(-1, 6)[positions].ravel()])
        # `bounds` (forwarded by the reference to least_squares: bundle_adjustment.py:301-313): scipy's checks and messages (least_squares.py);
        # infinite bounds everywhere = scipy's default = the unconstrained solver
        lohi = None
        if box is not None:
            from scipy.optimize import Bounds

            nx_all = 12 * n_cameras + 6 * all_use.size   # the bounds are those of the WHOLE parameter vector (every selected frame), on every rank
            if isinstance(box, Bounds):
                lb, ub = np.asarray(box.lb, dtype=np.float64), np.asarray(box.ub, dtype=np.float64)
            elif len(box) == 2:
                lb, ub = (np.asarray(b, dtype=np.float64) for b in box)
            else:
                raise ValueError("`bounds` must contain 2 elements.")
            lb, ub = (np.resize(lb, nx_all) if lb.ndim == 0 else lb), (np.resize(ub, nx_all) if ub.ndim == 0 else ub)
            if lb.shape != (nx_all,) or ub.shape != (nx_all,):
                raise ValueError("Inconsistent shapes between bounds and `x0`.")
            if np.any(lb >= ub):
                raise ValueError("Each lower bound must be strictly less than each upper bound.")
            x0_all = _X0.get() if not distributed else serialize_params(all_extrinsics, all_intrinsics, calib_poses[all_use])   # (every rank checks the whole start vector: they raise together)
            if np.any(x0_all < lb) or np.any(x0_all > ub):
                raise ValueError("Initial guess is outside of provided bounds")
            if np.isfinite(lb).any() or np.isfinite(ub).any():
                # this shard's part: the camera block and its own frames' blocks
                lohi = tuple(np.concatenate([b[: 12 * n_cameras], b[12 * n_cameras:].reshape(-1, 6)[positions].ravel()]) for b in (lb, ub))
        if use_frames.size == 0:
            # nothing to fit: scipy's least_squares on an empty residual vector returns x0 with status 1 (gtol) after one
            # evaluation (what the reference then returns: bundle_adjustment.py:307-327)
            x0 = _X0.get()
            result = OptimizeResult(x=x0, cost=0.0, fun=np.empty(0), jac=sp.csr_matrix((0, x0.size)), grad=np.zeros(x0.size), optimality=0.0,
                                    active_mask=np.zeros(x0.size), nfev=1, njev=1, status=1, message=solver.TERMINATION_MESSAGES[1], success=True)
            if kw["verbose"] >= 1:
                print(result.message)
                print("Function evaluations 1, initial cost 0.0000e+00, final cost 0.0000e+00, first-order optimality 0.00e+00.")
            ext, intr, poses = deserialize_params(x0, n_cameras)
            return ext, intr, poses, use_frames, result
        x0_on_device = False
        if local and not distributed and hasattr(prob_all, "lm_run") and use_frames.size == prob_all.F and bool((use_frames == np.arange(prob_all.F)).all()):
            # every frame, in order: the pre-filter's handle IS the problem (observations and start point are where they belong)
            prob, prob_all = prob_all, None
            prob.set_loss(kw["loss"], kw.get("f_scale", 1.0))
            x0_on_device = True
        elif local:        # the frames are already on the GPU (pre-filter): gather the selection there, no second upload
            prob, pre_sub = (pre_sub, None) if pre_sub is not None else (prob_all.subset(use_frames - lo, loss=kw["loss"], f_scale=kw.get("f_scale", 1.0)), None)
            x0_on_device = hasattr(prob, "lm_run")   # (mcba_create_subset gathers the parameters of the chosen frames as well)
        else:
            prob = ops.Problem(np.ascontiguousarray(all_calib_uvs[:, use_frames]), calib_objpoints, device=device, loss=kw["loss"], f_scale=kw.get("f_scale", 1.0), **pkw)
        if prob_all is not None:
            prob_all.close()
            prob_all = None
        # fix_intrinsics (BASELINE configs[1]): the library's 6-wide camera block -- role A of the linearisation alone, a 6C x 6C
        # camera system -- where it serves the rig (before anything allocates the solver buffers); otherwise flags on the 12-wide one
        free = None
        if fix_intrinsics and not (lohi is None and os.environ.get("MCBA_FIXED_COMPACT", "1") != "0" and hasattr(prob, "set_camera_block") and prob.set_camera_block(6)):
            free = np.tile(np.r_[np.zeros(6, bool), np.ones(6, bool)], n_cameras)
        comm = None
        if distributed:
            import torch

            comm = solver.make_comm(prob, torch.device(f"cuda:{device}"))
        # an explicit None disables that test, as scipy's check_tolerance does (least_squares.py: None -> 0); the device loop
        # treats a zero tolerance as "never satisfied".  A missing key keeps the reference's / scipy's defaults.
        tol = lambda name, default: 0.0 if kw.get(name, default) is None else float(kw.get(name, default))
        max_nfev = kw.get("max_nfev")
        if max_nfev is None:
            max_nfev = 100 * (12 * n_cameras + 6 * len(all_use))  # trf.py:437-438 on the GLOBAL vector: identical on every rank
        on_dev = x0_on_device and lohi is None
        result = solver.lm_solve(prob, None if on_dev else _X0.get(), ftol=tol("ftol", 1e-4), xtol=tol("xtol", 1e-8), gtol=tol("gtol", 1e-8),
                                 max_nfev=max_nfev, verbose=kw["verbose"] if rank == 0 else 0, free_cam_mask=free, comm=comm, x_scale=x_scale, x0_on_device=on_dev, lazy_grad=not distributed, bounds=lohi, **lm_kwargs)
        result = LazyOptimizeResult(result)

        # ---- OptimizeResult fields the reference's callers can rely on (trf.py:557-560).  `fun` and `jac` are LAZY: the residual
        # vector stays on the GPU as a detached buffer, and the robust-rescaled CSR Jacobian (116.6 M non-zeros = 1.4 GB on the host at
        # 6 x 10 000 x 54, 0.58 s to materialise, download and index) is produced when `result.jac` is first read -- from the handle,
        # which the result then keeps alive (observations + parameters, ~0.4 GB of HBM at that size) until the field has been
        # read or the result is dropped.  return_jac=False releases the handle at once.
        slot = result.lm["slot"]
        need_mask = not all_seen  # (every selected detection complete -- the pre-filter counted them on the GPU: no NaN mask to apply)
        shape4 = (n_cameras, use_frames.size, all_calib_uvs.shape[2], 2)
        # Which scalars are observed is fixed NOW, not when a lazy field is first read -- the caller may NaN out detections of its array in
        # place between this call and that read -- and it costs the call nothing: the detached residual vector carries NaN where
        # a scalar is missing (its own row mask), and `result.jac` takes its mask from the GPU's copy of the observations when it is produced
        # (ops.Problem.seen_bits; round 4 fetched those bits in every call: 0.1 ms at 6 x 10 000 x 54).
        vec = prob.residuals_detach(slot)
        own_mask = hasattr(prob, "seen_bits")   # (libmcba: NaN-marked vector; the CPU test double hands back zeros: mask from the caller's array)
        host_bits = None if (own_mask or not need_mask) else np.packbits(~np.isnan(all_calib_uvs)[:, use_frames])

        def fun(vec=vec, bits=host_bits, shape4=shape4, need_mask=need_mask, own_mask=own_mask):
            r = vec.download()
            if not need_mask:
                return r.ravel()
            return r[~np.isnan(r)] if own_mask else r[np.unpackbits(bits, count=r.size).astype(bool).reshape(shape4)]

        dict.__setitem__(result, "fun", _Lazy(fun))
        x_local = result.x.copy() if return_jac else None   # this process's frames (a frame-sharded run assembles the global vector below)
        grad = result.lm.pop("grad", None)   # (device-resident loop: packed next to x on the GPU and fetched with it -- mcba_lm_result)
        if grad is None:
            red = prob.get_reduced()
            gcam = np.zeros(12 * n_cameras)
            gcam[prob.cam_index if hasattr(prob, "cam_index") else slice(None)] = red["gc"]   # (6-wide camera block: the gradient entries of the intrinsics held fixed are 0, as with flags)
            grad = np.concatenate([gcam, prob.frame_gradient().ravel()])
            if free is not None:
                grad[: 12 * n_cameras][~free] = 0.0
        if distributed:
            # assemble the global vectors in selection order: cameras are identical on every rank, poses / frame gradients are gathered
            parts = [None] * world
            am = np.asarray(result.active_mask)
            dist.all_gather_object(parts, (positions, result.x[12 * n_cameras:].reshape(-1, 6), grad[12 * n_cameras:].reshape(-1, 6), am[12 * n_cameras:].reshape(-1, 6)))
            poses_all, gradf_all, amf_all = np.empty((all_use.size, 6)), np.empty((all_use.size, 6)), np.zeros((all_use.size, 6), dtype=am.dtype)
            for pos, po, gr, af in parts:
                poses_all[pos], gradf_all[pos], amf_all[pos] = po, gr, af
            result.x = np.concatenate([result.x[: 12 * n_cameras], poses_all.ravel()])
            grad = np.concatenate([grad[: 12 * n_cameras], gradf_all.ravel()])
            result.active_mask = np.concatenate([am[: 12 * n_cameras], amf_all.ravel()])   # (zeros without bounds; the cameras' entries are the same on every rank)
            if lohi is None:   # (bounded: the KKT residual the loop ended with -- gradient entries of the working set do not count)
                result.optimality = float(np.abs(grad).max())
            result.lm["frame_positions"] = positions
        if isinstance(grad, ops.DeviceArray):   # left on the GPU (mcba_lm_result): downloaded when `result.grad` is first read
            dict.__setitem__(result, "grad", _Lazy(grad.download))
        else:
            result.grad = grad
        if return_jac:
            # (no reference to the caller's array is kept: a result whose handle has to be released first copies its observations back from the GPU)
            jsrc = _JacobianSource(prob, calib_objpoints, x_local, slot, kw["loss"], kw.get("f_scale", 1.0), device, callable(kw["loss"]) or kw["loss"] != "linear", need_mask, shape4)
            dict.__setitem__(result, "jac", _Lazy(jsrc.csr))
            prob = None   # owned by the lazy `jac` field now (closed when that field is produced, with the result, or under MCBA_JAC_HOLD_MB pressure)
    finally:
        if prob_all is not None:
            prob_all.close()
        if pre_sub is not None:
            pre_sub.close()
        if prob is not None:
            prob.close()

    adjusted_extrinsics, adjusted_intrinsics, adjusted_calib_poses = deserialize_params(result.x, n_cameras)
    return adjusted_extrinsics, adjusted_intrinsics, adjusted_calib_poses, all_use, result


# BASELINE.json's north star calls the entry point `bundle_adjustment()`; the reference function is `bundle_adjust`.
bundle_adjustment = bundle_adjust
