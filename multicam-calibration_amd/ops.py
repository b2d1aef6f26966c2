"""ctypes binding of include/mcba.h (libmcba.so).  Thin: no arithmetic happens here.

The library is the ONLY compute path.  If it is missing or fails to load this module raises --
there is no CPU fallback (oracle/ is test infrastructure and is never imported from here)."""
import ctypes
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCBA_LIB") or os.path.join(PKG, "libmcba.so")   # MCBA_LIB: another build of the same ABI (development A/B runs)

LOSSES = {"linear": 0, "soft_l1": 1, "huber": 2, "cauchy": 3, "arctan": 4}
OK, ERR_HIP, ERR_ARG, ERR_NONFINITE, ERR_NODEVICE = 0, 1, 2, 3, 4

# every symbol include/mcba.h declares: (name, restype, argtypes)
_dp = ctypes.c_void_p   # double* (passed as the array's address: numpy's data_as(POINTER(c_double)) costs 2.8 us per argument, .ctypes.data 1.2 -- a call has 17 of them)
_ip = ctypes.POINTER(ctypes.c_int)
_h = ctypes.c_void_p
SYMBOLS = [
    ("mcba_abi_version", ctypes.c_int, []),
    ("mcba_last_error", ctypes.c_char_p, []),
    ("mcba_device_count", ctypes.c_int, [_ip]),
    ("mcba_create", ctypes.c_int, [ctypes.POINTER(_h), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("mcba_destroy", ctypes.c_int, [_h]),
    ("mcba_pool_trim", ctypes.c_int, []),
    ("mcba_device_bytes", ctypes.c_size_t, [_h]),
    ("mcba_trim", ctypes.c_int, [_h]),
    ("mcba_download_observations", ctypes.c_int, [_h, _dp]),
    ("mcba_set_x_scale", ctypes.c_int, [_h, _dp]),
    ("mcba_set_bounds", ctypes.c_int, [_h, _dp, _dp]),
    ("mcba_set_frozen", ctypes.c_int, [_h, ctypes.c_char_p]),
    ("mcba_residuals_detach", ctypes.c_int, [_h, ctypes.c_int, ctypes.POINTER(_h)]),
    ("mcba_buffer_count", ctypes.c_size_t, [_h]),
    ("mcba_buffer_download", ctypes.c_int, [_h, _dp]),
    ("mcba_buffer_free", ctypes.c_int, [_h]),
    ("mcba_error_histogram", ctypes.c_int, [_h, ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]),
    ("mcba_lm_fuse_status", ctypes.c_int, [_h, _dp, _ip]),
    ("mcba_set_strict_sync", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_get_strict_sync", ctypes.c_int, [_h]),
    ("mcba_set_stream", ctypes.c_int, [_h, ctypes.c_void_p]),
    ("mcba_upload_observations", ctypes.c_int, [_h, _dp, _dp]),
    ("mcba_set_loss", ctypes.c_int, [_h, ctypes.c_int, ctypes.c_double]),
    ("mcba_set_loss_table", ctypes.c_int, [_h, _dp]),
    ("mcba_set_trial", ctypes.c_int, [_h, _dp]),
    ("mcba_set_camera_block", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_get_camera_block", ctypes.c_int, [_h]),
    ("mcba_set_params", ctypes.c_int, [_h, ctypes.c_int, _dp]),
    ("mcba_get_params", ctypes.c_int, [_h, ctypes.c_int, _dp]),
    ("mcba_copy_params", ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int]),
    ("mcba_cost", ctypes.c_int, [_h, ctypes.c_int, _dp, _dp]),
    ("mcba_residuals", ctypes.c_int, [_h, ctypes.c_int, _dp]),
    ("mcba_seen_bits", ctypes.c_int, [_h, ctypes.POINTER(ctypes.c_ubyte)]),
    ("mcba_set_curvature_floor", ctypes.c_int, [_h, ctypes.c_double]),
    ("mcba_get_curvature_floor", ctypes.c_double, [_h]),
    ("mcba_jacobian_eval", ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int]),
    ("mcba_jacobian_download", ctypes.c_int, [_h, _dp, _dp]),
    ("mcba_linearize", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_build_reduced", ctypes.c_int, [_h, ctypes.c_double, ctypes.c_int]),
    ("mcba_reduced_size", ctypes.c_size_t, [_h]),
    ("mcba_bind_reduce_buffer", ctypes.c_int, [_h, ctypes.c_void_p]),
    ("mcba_get_reduced", ctypes.c_int, [_h, _dp]),
    ("mcba_step", ctypes.c_int, [_h, _dp, ctypes.c_double, ctypes.c_int, ctypes.c_int]),
    ("mcba_step_linearize", ctypes.c_int, [_h, _dp, ctypes.c_double, ctypes.c_int, ctypes.c_int]),
    ("mcba_accept_linearization", ctypes.c_int, [_h]),
    ("mcba_get_trial", ctypes.c_int, [_h, _dp]),
    ("mcba_reduce_fetch", ctypes.c_int, [_h, ctypes.c_double, ctypes.c_int, _dp]),
    ("mcba_step_fetch", ctypes.c_int, [_h, _dp, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp]),
    ("mcba_lm_set_state", ctypes.c_int, [_h, _dp]),
    ("mcba_lm_trial", ctypes.c_int, [_h, _dp]),
    ("mcba_lm_decide_reduce", ctypes.c_int, [_h, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int]),
    ("mcba_lm_rebuild", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_lm_fetch", ctypes.c_int, [_h, _dp]),
    ("mcba_lm_iterate", ctypes.c_int, [_h, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp]),
    ("mcba_lm_auto_config", ctypes.c_int, [_h, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_char_p]),
    ("mcba_lm_auto_solve", ctypes.c_int, [_h, ctypes.c_ulonglong, ctypes.c_int]),
    ("mcba_lm_set_decrease_floor", ctypes.c_int, [_h, ctypes.c_double]),
    ("mcba_lm_auto_trial", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_lm_auto_reduce", ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int]),
    ("mcba_lm_auto_tick", ctypes.c_int, [_h, ctypes.c_ulonglong, ctypes.c_int]),
    ("mcba_lm_auto_wait", ctypes.c_int, [_h, ctypes.c_ulonglong, _dp]),
    ("mcba_get_cam_step", ctypes.c_int, [_h, _dp]),
    ("mcba_comm_unique_id", ctypes.c_int, [ctypes.c_char_p]),
    ("mcba_comm_init", ctypes.c_int, [_h, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    ("mcba_comm_allreduce", ctypes.c_int, [_h, ctypes.c_size_t, ctypes.c_size_t]),
    ("mcba_comm_count", ctypes.c_int, [_h, _ip]),
    ("mcba_comm_destroy", ctypes.c_int, [_h]),
    ("mcba_get_frame_gradient", ctypes.c_int, [_h, _dp]),
    ("mcba_frame_errors", ctypes.c_int, [_h, ctypes.c_int, _dp, _dp]),
    ("mcba_error_median", ctypes.c_int, [_h, ctypes.c_char_p, _dp, _dp]),
    ("mcba_create_subset", ctypes.c_int, [ctypes.POINTER(_h), _h, _ip, ctypes.c_int]),
    ("mcba_create_views", ctypes.c_int, [ctypes.POINTER(_h), _h, _ip, ctypes.c_int]),
    ("mcba_calib_complete", ctypes.c_int, [_h, ctypes.POINTER(ctypes.c_ubyte)]),
    ("mcba_calib_homographies", ctypes.c_int, [_h, _ip, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte)]),
    ("mcba_calib_view_poses", ctypes.c_int, [_h, _ip, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte)]),
    ("mcba_calib_graph", ctypes.c_int, [_h, _ip, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp]),
    ("mcba_calib_start", ctypes.c_int, [_h, _ip, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte), _dp, ctypes.POINTER(ctypes.c_ubyte)]),
    ("mcba_calib_poses", ctypes.c_int, [_h, _dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_ubyte)]),
    ("mcba_calib_pairwise", ctypes.c_int, [_h, _ip, ctypes.c_int, _dp, _dp]),
    ("mcba_calib_consensus", ctypes.c_int, [_h, _dp, _dp]),
    ("mcba_pose_pairwise", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _ip, ctypes.c_int, ctypes.c_int, _dp, _dp]),
    ("mcba_pose_consensus", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp]),
    ("mcba_prefilter", ctypes.c_int, [_h, _dp, _dp, _dp, ctypes.c_double, ctypes.POINTER(ctypes.c_ubyte), _dp]),
    ("mcba_prefilter_subset", ctypes.c_int, [_h, _dp, _dp, _dp, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte), _dp, ctypes.POINTER(_h)]),
    ("mcba_lm_run", ctypes.c_int, [_h, _dp, _dp, ctypes.c_char_p, _dp]),
    ("mcba_lm_history", ctypes.c_int, [_h, _dp, ctypes.c_size_t]),
    ("mcba_lm_result", ctypes.c_int, [_h, ctypes.c_int, _dp, _dp, ctypes.POINTER(_h)]),
    ("mcba_undistort_points", ctypes.c_int, [ctypes.c_size_t, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp]),
    ("mcba_calib_normal_equations", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.c_int, _dp]),
    ("mcba_reprojection_diagnostics", ctypes.c_int, [_h, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp, _dp]),
    ("mcba_triangulate", ctypes.c_int, [ctypes.c_int, ctypes.c_size_t, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp]),
    ("mcba_profile_enable", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_profile_stride", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_profile_read", ctypes.c_int, [_h, _dp, _ip, ctypes.c_int, _ip]),
    ("mcba_profile_bracket_overhead", ctypes.c_int, [_h, ctypes.c_int, _dp]),
    ("mcba_profile_exact", ctypes.c_int, [_h, ctypes.c_int]),
    ("mcba_profile_names", ctypes.c_char_p, []),
    ("mcba_synchronize", ctypes.c_int, [_h]),
    ("mcba_fp64_issue_rate", ctypes.c_int, [ctypes.c_int, _dp]),
]

LM_STATE = 32  # MCBA_LM_STATE of include/mcba.h

_lib = None


class McbaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmcba error {code}: {msg}")
        self.code = code


def load_library():
    """Load libmcba.so once.  torch is imported first so both share one HIP runtime (same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of this package. "
            "Build it with `python -m multicam_calibration_amd.build` (needs hipcc)."
        )
    try:
        import torch  # noqa: F401  (plumbing: device memory for collectives, streams, torch.distributed)
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceArray:
    """A float64 array that lives on the GPU and outlives the Problem that produced it (mcba_residuals_detach):
    `download()` copies it to the host once and releases the device buffer; dropping the object releases it unread."""

    def __init__(self, lib, handle, shape):
        self.lib, self.handle, self.shape = lib, handle, tuple(shape)
        self._host = None

    def download(self):
        if self._host is None:
            out = np.empty(self.shape)
            rc = self.lib.mcba_buffer_download(self.handle, _p(out))
            if rc != OK:
                raise McbaError(rc, self.lib.mcba_last_error().decode())
            self._host = out
            self.free()
        return self._host

    def free(self):
        if self.handle:
            self.lib.mcba_buffer_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def fp64_issue_rate(device=0):
    """TFLOP/s of independent v_fma_f64 the device sustains at one wavefront per SIMD (include/mcba.h: mcba_fp64_issue_rate)."""
    lib = load_library()
    t = ctypes.c_double()
    rc = lib.mcba_fp64_issue_rate(int(device), ctypes.byref(t))
    if rc != OK:
        raise McbaError(rc, lib.mcba_last_error().decode())
    return t.value


def pool_trim():
    """Return every parked device / pinned buffer of destroyed handles to the driver (include/mcba.h: mcba_pool_trim)."""
    load_library().mcba_pool_trim()


class Problem:
    """One handle = one GPU = this process's shard of frames."""

    def __init__(self, uvs, objpoints, device=0, loss="soft_l1", f_scale=1.0, stream=None, upload=True):
        """upload=False: the handle is created, the observations are NOT sent yet -- `prefilter` uploads and scores them in one call."""
        self.lib = load_library()
        uvs = _f64(uvs)
        objpoints = _f64(objpoints)
        if uvs.ndim != 4 or uvs.shape[3] != 2 or objpoints.shape != (uvs.shape[2], 3):
            raise ValueError("uvs must be (C,F,N,2) and objpoints (N,3)")
        self.C, self.F, self.N = uvs.shape[:3]
        self.cw = 12
        self.n = 12 * self.C
        self.nx = 12 * self.C + 6 * self.F
        self.handle = _h()
        self._chk(self.lib.mcba_create(ctypes.byref(self.handle), self.C, self.F, self.N, int(device)))
        if stream is not None:
            self._chk(self.lib.mcba_set_stream(self.handle, ctypes.c_void_p(int(stream))))
        if upload:
            self._chk(self.lib.mcba_upload_observations(self.handle, _p(uvs), _p(objpoints)))
        else:
            self._pending = (uvs, objpoints)
        self.set_loss(loss, f_scale)

    def prefilter(self, x, outlier_threshold=None):
        """The reference's pre-filter (bundle_adjustment.py:265-285) in one C-ABI crossing and one host synchronisation
        (include/mcba.h: mcba_prefilter): the observations (if the handle was created with upload=False) and the parameters x of every
        frame go up, the selection comes back.  Returns (status (F,) uint8 -- bit 0 used, bit 1 excluded as an outlier, bit 2 complete in
        every camera --, threshold, info (8,))."""
        x = _f64(x)
        if x.shape != (self.nx,):
            raise ValueError(f"x must have {self.nx} entries")
        status = np.empty(self.F, np.uint8)
        info = np.empty(8)
        uvs, obj = getattr(self, "_pending", None) or (None, None)
        thr = float("nan") if outlier_threshold is None else float(outlier_threshold)
        self._chk(self.lib.mcba_prefilter(self.handle, None if uvs is None else _p(uvs), None if obj is None else _p(obj), _p(x), thr,
                                          status.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _p(info)))
        self._pending = None
        return status, float(info[0]), info

    _HOST_VIEWS = frozenset(("nsys", "_all", "_red", "_trial", "_state", "_all_p", "_red_p", "_trial_p", "_red_views", "_dc", "_dc_p"))

    def __getattr__(self, name):
        # the host-side images of the reduce buffer / trial scalars / LM state are made when first touched: the handle that only pre-filters
        # (api.select_frames) never needs them
        if name in Problem._HOST_VIEWS and "n" in self.__dict__:
            self._init_host_views()
            return self.__dict__[name]
        raise AttributeError(name)

    def prefilter_subset(self, x, outlier_threshold=None, n_frames=None, loss=None, f_scale=None):
        """`prefilter` + the gather of the kept frames in the same crossing when no random draw stands in between (include/mcba.h:
        mcba_prefilter_subset).  Returns (status, threshold, info, sub): info[7] = 0 nothing kept, 1 the caller must draw its subsample
        (sub None), 2 every frame kept in order (sub None: solve on this handle), 3 sub = a new Problem holding the kept frames in order."""
        x = _f64(x)
        if x.shape != (self.nx,):
            raise ValueError(f"x must have {self.nx} entries")
        status = np.empty(self.F, np.uint8)
        info = np.empty(8)
        uvs, obj = getattr(self, "_pending", None) or (None, None)
        thr = float("nan") if outlier_threshold is None else float(outlier_threshold)
        sub = _h()
        self._chk(self.lib.mcba_prefilter_subset(self.handle, None if uvs is None else _p(uvs), None if obj is None else _p(obj), _p(x), thr, -1 if n_frames is None else int(n_frames),
                                                 status.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _p(info), ctypes.byref(sub)))
        self._pending = None
        new = None
        if sub:
            new = Problem.__new__(Problem)
            new.lib = self.lib
            new.C, new.N, new.F = self.C, self.N, int(info[4] - info[5])
            new.cw, new.n = 12, 12 * self.C
            new.nx = 12 * new.C + 6 * new.F
            new.handle = sub
            if loss is not None:
                new.set_loss(loss, 1.0 if f_scale is None else f_scale)
        return status, float(info[0]), info, new

    def _init_host_views(self):
        self.nsys = self.n * self.n + 3 * self.n + 16
        self._all = np.zeros(self.nsys + 8 + LM_STATE)   # system | trial scalars | LM state, as the device lays them out
        self._red = self._all[: self.nsys]
        self._trial = self._all[self.nsys : self.nsys + 8]
        self._state = self._all[self.nsys + 8 :]
        self._all_p, self._red_p, self._trial_p = _p(self._all), _p(self._red), _p(self._trial)
        self._red_views = self.split_reduced(self._red)
        self._dc = np.empty(self.n)
        self._dc_p = _p(self._dc)

    # ---- frame pre-filter on the GPU + frame subsets without a second upload (bundle_adjustment.py:265-298)
    def frame_errors(self, slot):
        """(mean_cf, full_cf), both (C,F): nan-mean reprojection error per (camera, frame) and number of complete points."""
        mean = np.empty((self.C, self.F))
        full = np.empty((self.C, self.F))
        self._chk(self.lib.mcba_frame_errors(self.handle, slot, _p(mean), _p(full)))
        return mean, full

    def error_median(self, frame_mask=None):
        """np.nanmedian of the per-point errors of the last frame_errors() over the masked frames -> (median, count)."""
        med, cnt = ctypes.c_double(), ctypes.c_double()
        m = None if frame_mask is None else np.ascontiguousarray(frame_mask, dtype=np.uint8).tobytes()
        self._chk(self.lib.mcba_error_median(self.handle, m, ctypes.byref(med), ctypes.byref(cnt)))
        return med.value, int(cnt.value)

    def error_histogram(self, frame_mask, prefix, pass_):
        """One pass of the radix select over THIS handle's errors (frame-sharded pre-filter): 256 counts (uint64) of the next byte
        among the values whose leading `pass_` bytes equal `prefix`; frame_mask None = the mask of the previous call."""
        hist = np.zeros(256, dtype=np.uint64)
        m = None if frame_mask is None else np.ascontiguousarray(frame_mask, dtype=np.uint8).tobytes()
        self._chk(self.lib.mcba_error_histogram(self.handle, m, int(prefix), int(pass_), hist.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))))
        return hist

    def subset(self, frames, loss=None, f_scale=None):
        """A new Problem holding the observations of `frames` (indices into this one), gathered on the GPU."""
        idx = np.ascontiguousarray(frames, dtype=np.int32)
        new = Problem.__new__(Problem)
        new.lib = self.lib
        new.C, new.N, new.F = self.C, self.N, int(idx.size)
        new.cw, new.n = 12, 12 * self.C
        new.nx = 12 * new.C + 6 * new.F
        new.handle = _h()
        self._chk(self.lib.mcba_create_subset(ctypes.byref(new.handle), self.handle, idx.ctypes.data_as(_ip), int(idx.size)))
        if loss is not None:
            new.set_loss(loss, 1.0 if f_scale is None else f_scale)
        return new

    # ---- calibrate() on the device (include/mcba.h: the mcba_calib_* block; reference calibration.py:11-277)
    def _sibling(self, handle, F):
        new = Problem.__new__(Problem)
        new.lib = self.lib
        new.C, new.N, new.F = self.C, self.N, int(F)
        new.cw, new.n = 12, 12 * self.C
        new.nx = 12 * new.C + 6 * new.F
        new.handle = handle
        return new

    def view_subset(self, views, loss=None, f_scale=None):
        """A new Problem of C cameras x len(views) frames: frame j holds the detection of view j = (camera, frame) in its camera alone."""
        v = np.ascontiguousarray(views, dtype=np.int32).reshape(-1, 2)
        sub = _h()
        self._chk(self.lib.mcba_create_views(ctypes.byref(sub), self.handle, v.ctypes.data_as(_ip), len(v)))
        new = self._sibling(sub, len(v))
        if loss is not None:
            new.set_loss(loss, 1.0 if f_scale is None else f_scale)
        return new

    def calib_complete(self):
        """(C,F) bool: every scalar of the detection present."""
        out = np.empty((self.C, self.F), np.uint8)
        self._chk(self.lib.mcba_calib_complete(self.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))))
        return out.view(np.bool_)

    def calib_homographies(self, views):
        """(V,3,3) board-plane -> pixel homographies of the (camera, frame) views, H[2,2] = 1; NaN for incomplete views."""
        v = np.ascontiguousarray(views, dtype=np.int32).reshape(-1, 2)
        H = np.empty((len(v), 3, 3))
        self._chk(self.lib.mcba_calib_homographies(self.handle, v.ctypes.data_as(_ip), len(v), _p(H), None))
        return H

    def calib_view_poses(self, views, intr9, undistort_iterations=8, max_evaluations=60):
        """(V,6) board poses of the listed views with the cameras' intrinsics intr9 (C,9); NaN rows where none came out."""
        v = np.ascontiguousarray(views, dtype=np.int32).reshape(-1, 2)
        k = _f64(intr9).reshape(self.C, 9)
        out = np.empty((len(v), 6))
        self._chk(self.lib.mcba_calib_view_poses(self.handle, v.ctypes.data_as(_ip), len(v), _p(k), int(undistort_iterations), int(max_evaluations), _p(out), None))
        return out

    def calib_start(self, views, image_sizes, undistort_iterations=8, max_evaluations=60, want_closed=False):
        """The closed-form start of every camera from its sampled views, one crossing (include/mcba.h: mcba_calib_start): homographies, Zhang's K
        per camera, the views' poses with it.  image_sizes (C,2) = (width, height).  Returns (k4 (C,4) fx fy cx cy, poses (V,6))
        [+ closed (C,) bool: the closed form was used, not the fallback]."""
        v = np.ascontiguousarray(views, dtype=np.int32).reshape(-1, 2)
        sz = _f64(image_sizes).reshape(self.C, 2)
        k4, poses = np.empty((self.C, 4)), np.empty((len(v), 6))
        closed = np.empty(self.C, np.uint8)
        self._chk(self.lib.mcba_calib_start(self.handle, v.ctypes.data_as(_ip), len(v), _p(sz), int(undistort_iterations), int(max_evaluations), _p(k4),
                                            closed.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _p(poses), None))
        return (k4, poses, closed.astype(bool)) if want_closed else (k4, poses)

    def calib_poses(self, intr9, undistort_iterations=8, max_evaluations=60, want_poses=False, want_evals=False):
        """Board pose of every (camera, frame) with a complete detection, left on the device for calib_pairwise / calib_consensus.
        Returns (ok (C,F) bool, poses (C,F,6) or None, evaluations (C,F) uint8 or None)."""
        k = _f64(intr9).reshape(self.C, 9)
        ok = np.empty((self.C, self.F), np.uint8)
        poses = np.empty((self.C, self.F, 6)) if want_poses else None
        ev = np.empty((self.C, self.F), np.uint8) if want_evals else None
        ub = ctypes.POINTER(ctypes.c_ubyte)
        self._chk(self.lib.mcba_calib_poses(self.handle, _p(k), int(undistort_iterations), int(max_evaluations), None if poses is None else _p(poses), ok.ctypes.data_as(ub),
                                            None if ev is None else ev.ctypes.data_as(ub)))
        return ok.view(np.bool_), poses, ev

    def calib_pairwise(self, edges):
        """(E,6) median relative transforms T2 T1^-1 of the camera pairs `edges` (E,2), and the number of frames each pair shares."""
        e = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        out, cnt = np.empty((len(e), 6)), np.empty(len(e))
        self._chk(self.lib.mcba_calib_pairwise(self.handle, e.ctypes.data_as(_ip), len(e), _p(out), _p(cnt)))
        return out, cnt

    def calib_graph(self, tree, root, want_transforms=False):
        """The pose graph in one crossing (include/mcba.h: mcba_calib_graph): (extrinsics (C,6) chained down `tree` from `root`, consensus board
        poses (F,6)) [+ the tree edges' median transforms (E,6) and shared-frame counts (E,)]."""
        e = np.ascontiguousarray(tree, dtype=np.int32).reshape(-1, 2)
        ext, poses = np.empty((self.C, 6)), np.empty((self.F, 6))
        tr, cnt = np.empty((len(e), 6)), np.empty(len(e))
        self._chk(self.lib.mcba_calib_graph(self.handle, e.ctypes.data_as(_ip), len(e), int(root), _p(ext), _p(poses), _p(tr), _p(cnt)))
        return (ext, poses, tr, cnt) if want_transforms else (ext, poses)

    def calib_consensus(self, extrinsics):
        """(F,6) consensus board poses in world coordinates for the extrinsics (C,6)."""
        ext = _f64(extrinsics).reshape(self.C, 6)
        out = np.empty((self.F, 6))
        self._chk(self.lib.mcba_calib_consensus(self.handle, _p(ext), _p(out)))
        return out

    def reprojection_diagnostics(self, slot, dist5=None, undistort_iterations=5, arrays=True):
        """(median_error (C,), reprojections (C,F,N,2), transformed_reprojections (C,F,N,2)) as plot_residuals computes them
        (viz.py:166-186); the two arrays are None when arrays=False."""
        med = np.empty(self.C)
        rep = np.empty((self.C, self.F, self.N, 2)) if arrays else None
        tra = np.empty((self.C, self.F, self.N, 2)) if arrays else None
        d5 = None if dist5 is None else _f64(dist5).reshape(self.C, 5)
        self._chk(self.lib.mcba_reprojection_diagnostics(self.handle, slot, None if d5 is None else _p(d5), int(undistort_iterations), _p(med),
                                                         None if rep is None else _p(rep), None if tra is None else _p(tra)))
        return med, rep, tra

    def _chk(self, rc):
        if rc != OK:
            raise McbaError(rc, self.lib.mcba_last_error().decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mcba_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self):
        """Bytes of device memory this handle holds right now (include/mcba.h: mcba_device_bytes)."""
        return int(self.lib.mcba_device_bytes(self.handle))

    def trim(self):
        """Give back every device buffer except the observations, the board and the parameter slots (include/mcba.h: mcba_trim)."""
        self._chk(self.lib.mcba_trim(self.handle))

    def download_observations(self):
        """(C,F,N,2): the observations as this handle holds them (include/mcba.h: mcba_download_observations)."""
        out = np.empty((self.C, self.F, self.N, 2))
        self._chk(self.lib.mcba_download_observations(self.handle, _p(out)))
        return out

    def set_camera_block(self, width):
        """Camera block width (include/mcba.h: mcba_set_camera_block): 12 = every camera parameter is a variable (the reference);
        6 = the intrinsics of every camera are held fixed (BASELINE configs[1]) -- the camera system, the camera step and every other
        camera-system vector then have 6 entries per camera (rho, t).  Before the first solver call of the handle.  Returns False
        (and changes nothing) if the library declines the width for this rig (more than 26 cameras: hold them with flags instead)."""
        rc = self.lib.mcba_set_camera_block(self.handle, int(width))
        if rc == ERR_ARG and int(width) == 6 and "26 cameras" in self.lib.mcba_last_error().decode():
            return False
        self._chk(rc)
        self.cw = int(width)
        self.n = self.cw * self.C
        self._init_host_views()
        return True

    @property
    def cam_index(self):
        """Positions in the parameter vector x of the camera system's variables, in the system's order."""
        if self.cw == 12:
            return np.arange(12 * self.C)
        return (12 * np.arange(self.C)[:, None] + 6 + np.arange(6)[None, :]).ravel()

    _loss_fn = None   # least_squares' callable `loss`, if that is what set_loss was given

    def set_loss(self, loss, f_scale=1.0):
        """One of least_squares' five names, or its CALLABLE form rho(z) -> array (3, m) = (rho, rho', rho'') (least_squares.py:160-227; the
        reference forwards `loss` untouched: bundle_adjustment.py:301-313).  A callable is evaluated HERE, on the residuals the GPU computed, each
        time a point is linearised or a trial cost is asked for; its values go to the GPU as a table (include/mcba.h: mcba_set_loss_table).
        Host-driven loop only (solver.py)."""
        if callable(loss):
            if not f_scale > 0:
                raise ValueError("`f_scale` must be positive.")
            self._loss_fn, self._loss_fs, self._loss_valid = loss, float(f_scale), None
            self._loss_forget()
            return
        if loss not in LOSSES:
            raise ValueError(f"loss must be one of {sorted(LOSSES)} or a callable")
        self._loss_fn = None
        self._chk(self.lib.mcba_set_loss(self.handle, LOSSES[loss], float(f_scale)))

    @property
    def loss_is_callable(self):
        return self._loss_fn is not None

    def loss_values(self, slot):
        """The caller's rho at the residuals of x[slot], as scipy's loss_function wrapper hands it to the solver (least_squares.py: construct_loss_function):
        (valid (C,F,N,2) bool, z (m,), rho (3, m)) over the observed scalars in the reference's row order; rho is None if a residual is not finite.
        The function is called ONCE per point: a trial point's cost and, if the step is accepted, its table come from the same evaluation
        (the values are kept per parameter slot until that slot is written again)."""
        hit = self._loss_cache.get(slot) if self._loss_cache is not None else None
        if hit is not None:
            return hit
        if self._loss_valid is None:
            self._loss_valid = np.unpackbits(self.seen_bits(), count=2 * self.C * self.F * self.N).astype(bool).reshape(self.C, self.F, self.N, 2)
            self._loss_index = None if self._loss_valid.all() else np.flatnonzero(self._loss_valid)   # (a complete recording needs no gather / scatter)
        # (host buffers of the callable path are kept with the handle: a fresh 52 / 156 MB numpy array per evaluation costs more in page faults
        #  than the copy that fills it)
        if self._loss_res_buf is None:
            self._loss_res_buf = np.empty((self.C, self.F, self.N, 2))
        self._chk(self.lib.mcba_residuals(self.handle, slot, _p(self._loss_res_buf)))
        f = self._loss_res_buf.reshape(-1)
        f = f.take(self._loss_index) if self._loss_index is not None else f.copy()
        if not np.isfinite(f).all():
            out = (self._loss_valid, None, None)
        else:
            if self._loss_fs != 1.0:
                f /= self._loss_fs
            z = np.square(f, out=f)
            rho = np.asarray(self._loss_fn(z), dtype=np.float64)
            if rho.shape != (3, z.size):
                raise ValueError("The return value of `loss` callable has wrong shape.")
            out = (self._loss_valid, z, rho)
        if self._loss_cache is None:
            self._loss_cache = {}
        self._loss_cache[slot] = out
        return out

    _loss_cache = None
    _loss_res_buf = None
    _loss_tab_buf = None

    def _loss_forget(self, slot=None):
        """x[slot] is about to change (None: every slot): what the caller's function said about it no longer holds."""
        if self._loss_cache:
            if slot is None:
                self._loss_cache.clear()
            else:
                self._loss_cache.pop(slot, None)

    def _callable_cost(self, slot):
        _, _, rho = self.loss_values(slot)
        return np.inf if rho is None else 0.5 * self._loss_fs**2 * float(np.sum(rho[0]))

    def _refresh_loss_table(self, slot):
        valid, z, rho = self.loss_values(slot)
        if rho is None:
            raise ValueError("Residuals are not finite in the initial point.")
        # three planes over every scalar of the observation array: 0.5 f_scale^2 rho, rho', scipy's J_scale^2 = max(rho' + 2 rho'' z, EPS)
        # (common.py:720-731; rho'' / f_scale^2 * f^2 = rho'' z); 0 where nothing is observed
        js2 = rho[2] * z
        js2 *= 2.0
        js2 += rho[1]
        np.maximum(js2, np.finfo(float).eps, out=js2)
        if self._loss_tab_buf is None:
            self._loss_tab_buf = np.zeros((3, valid.size))   # (the unobserved scalars stay 0 for good: only the observed ones are ever written)
        tab = self._loss_tab_buf
        if self._loss_index is None:
            np.multiply(rho[0], 0.5 * self._loss_fs**2, out=tab[0])
            tab[1], tab[2] = rho[1], js2
        else:
            tab[0][self._loss_index] = 0.5 * self._loss_fs**2 * rho[0]
            tab[1][self._loss_index] = rho[1]
            tab[2][self._loss_index] = js2
        self._chk(self.lib.mcba_set_loss_table(self.handle, _p(tab)))

    def set_params(self, slot, x):
        x = _f64(x)
        if x.shape != (self.nx,):
            raise ValueError(f"x must have {self.nx} entries")
        self._loss_forget(slot)
        self._chk(self.lib.mcba_set_params(self.handle, slot, _p(x)))

    def get_params(self, slot):
        x = np.empty(self.nx)
        self._chk(self.lib.mcba_get_params(self.handle, slot, _p(x)))
        return x

    def copy_params(self, dst, src):
        self._loss_forget(dst)
        self._chk(self.lib.mcba_copy_params(self.handle, dst, src))

    def cost(self, slot):
        c, n = ctypes.c_double(), ctypes.c_double()
        self._chk(self.lib.mcba_cost(self.handle, slot, ctypes.byref(c), ctypes.byref(n)))
        if self._loss_fn is not None:
            return self._callable_cost(slot), n.value
        return c.value, n.value

    def residuals(self, slot):
        """(C,F,N,2) observed - predicted, 0 where the observation is missing."""
        r = np.empty((self.C, self.F, self.N, 2))
        self._chk(self.lib.mcba_residuals(self.handle, slot, _p(r)))
        return r

    def residuals_detach(self, slot):
        """The same array left on the GPU as a DeviceArray (no device-to-host copy until `.download()`)."""
        buf = _h()
        self._chk(self.lib.mcba_residuals_detach(self.handle, slot, ctypes.byref(buf)))
        return DeviceArray(self.lib, buf, (self.C, self.F, self.N, 2))

    def set_curvature_floor(self, floor):
        """Curvature weight of the linearisations enqueued from now on: max(Triggs, floor * rho') -- 1 = IRLS, 0.1 = Triggs with a
        floor (include/mcba.h: mcba_set_curvature_floor).  Returns the value in force before."""
        old = self.lib.mcba_get_curvature_floor(self.handle)
        self._chk(self.lib.mcba_set_curvature_floor(self.handle, float(floor)))
        return old

    def seen_bits(self):
        """numpy.packbits(~numpy.isnan(uvs)) of the uploaded (C,F,N,2) observations, computed on the GPU from its own copy
        (mcba_seen_bits): the row selection of the reference's residual vector / Jacobian (bundle_adjustment.py:68-69)."""
        bits = np.empty((2 * self.C * self.F * self.N + 7) // 8, np.uint8)
        self._chk(self.lib.mcba_seen_bits(self.handle, bits.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))))
        return bits

    def set_x_scale(self, x_scale):
        """least_squares' numeric x_scale for this shard's parameter vector (nx positive numbers), or None for 'jac'."""
        if x_scale is None:
            self._chk(self.lib.mcba_set_x_scale(self.handle, None))
            return
        xs = _f64(x_scale)
        if xs.shape != (self.nx,):
            raise ValueError("Inconsistent shapes between `x_scale` and `x0`.")
        rc = self.lib.mcba_set_x_scale(self.handle, _p(xs))
        if rc == ERR_ARG:
            raise ValueError(self.lib.mcba_last_error().decode())
        self._chk(rc)

    def set_strict_sync(self, on=True):
        """on (the default of a new handle): the fused back-substitution's readers acquire the solve's release word with an agent-scope fence,
        the form the HIP memory model asks for; off: relaxed loads relying on gfx950's ordering (include/mcba.h: mcba_set_strict_sync).
        Returns the setting in force before."""
        old = bool(self.lib.mcba_get_strict_sync(self.handle))
        self._chk(self.lib.mcba_set_strict_sync(self.handle, int(bool(on))))
        return old

    def set_bounds(self, lo, hi):
        """Box constraints lo <= x <= hi in the layout of x (None, None: none): trial points of step / step_linearize / step_fetch are projected
        onto the box (include/mcba.h: mcba_set_bounds)."""
        if lo is None:
            self._chk(self.lib.mcba_set_bounds(self.handle, None, None))
            return
        lo, hi = _f64(lo), _f64(hi)
        if lo.shape != (self.nx,) or hi.shape != (self.nx,):
            raise ValueError("Inconsistent shapes between bounds and `x0`.")
        rc = self.lib.mcba_set_bounds(self.handle, _p(lo), _p(hi))
        if rc == ERR_ARG:
            raise ValueError(self.lib.mcba_last_error().decode())
        self._chk(rc)

    def set_frozen(self, mask):
        """Frame coordinates taken out of the next linear solves (the working set of the bounded loop; include/mcba.h: mcba_set_frozen)."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).tobytes()
        if m is not None and len(m) != self.nx:
            raise ValueError(f"mask must have {self.nx} entries")
        self._chk(self.lib.mcba_set_frozen(self.handle, m))

    def fuse_status(self):
        """(number of the last tick whose fused back-substitution gave up waiting for the solve -- 0: never --, fused launch still in use)."""
        t, f = ctypes.c_double(), ctypes.c_int()
        self._chk(self.lib.mcba_lm_fuse_status(self.handle, ctypes.byref(t), ctypes.byref(f)))
        return t.value, bool(f.value)

    def jacobian_eval(self, slot, robust_scaled=False):
        self._chk(self.lib.mcba_jacobian_eval(self.handle, slot, int(bool(robust_scaled))))

    def jacobian_download(self, want_res=True):
        jac = np.empty((self.C, self.F, self.N, 2, 18))
        res = np.empty((self.C, self.F, self.N, 2)) if want_res else None
        self._chk(self.lib.mcba_jacobian_download(self.handle, _p(jac), _p(res) if want_res else None))
        return jac, res

    def linearize(self, slot):
        if self._loss_fn is not None:
            self._refresh_loss_table(slot)   # the caller's rho at THIS point
        self._chk(self.lib.mcba_linearize(self.handle, slot))

    def build_reduced(self, lam, rank_slot=0):
        self._chk(self.lib.mcba_build_reduced(self.handle, float(lam), int(rank_slot)))

    def reduced_size(self):
        return int(self.lib.mcba_reduced_size(self.handle))

    def bind_reduce_buffer(self, device_ptr):
        self._chk(self.lib.mcba_bind_reduce_buffer(self.handle, ctypes.c_void_p(int(device_ptr) if device_ptr else None)))

    def get_reduced(self):
        """dict(S0 (n,n), rhs, diagU, gc, scal(16)) of this shard (or of all shards after an all-reduce)."""
        self._chk(self.lib.mcba_get_reduced(self.handle, _p(self._red)))
        return self.split_reduced(self._red)

    def split_reduced(self, r):
        n = self.n
        return dict(S0=r[: n * n].reshape(n, n), rhs=r[n * n : n * n + n], diagU=r[n * n + n : n * n + 2 * n], gc=r[n * n + 2 * n : n * n + 3 * n], scal=r[n * n + 3 * n : n * n + 3 * n + 16])

    def reduce_fetch(self, lam, rank_slot=0):
        """build_reduced + get_reduced in one ABI crossing.  The returned arrays are VIEWS of a buffer that the
        next call overwrites (the LM driver adds its damping in place)."""
        rc = self.lib.mcba_reduce_fetch(self.handle, lam, rank_slot, self._red_p)
        if rc:
            self._chk(rc)
        return self._red_views

    def step_fetch(self, delta_cam, lam, src, dst, linearize):
        """step / step_linearize + get_trial in one ABI crossing; returns a view of the 8 trial scalars."""
        self._dc[:] = delta_cam
        if self._loss_fn is not None and linearize:
            raise ValueError("a callable loss cannot linearise its trial point speculatively (its table is not known before the step): use speculative=False")
        self._loss_forget(dst)
        rc = self.lib.mcba_step_fetch(self.handle, self._dc_p, lam, src, dst, 1 if linearize else 0, self._trial_p)
        if rc:
            self._chk(rc)
        if self._loss_fn is not None:
            self._trial[0] = self._callable_cost(dst)   # the trial cost is the caller's function on the trial residuals
        return self._trial

    def step(self, delta_cam, lam, src, dst):
        d = _f64(delta_cam)
        self._loss_forget(dst)
        self._chk(self.lib.mcba_step(self.handle, _p(d), float(lam), src, dst))
        if self._loss_fn is not None:
            # the trial cost of THIS shard's frames is the caller's function on their residuals: put in place of the kernel's BEFORE the all-reduce
            # of the trial scalars (mcba_residuals reuses the scalars' slots: the step's other scalars are fetched first and written back)
            t = self.get_trial()
            t[0] = self._callable_cost(dst)
            self._chk(self.lib.mcba_set_trial(self.handle, _p(t)))

    def step_linearize(self, delta_cam, lam, src, dst):
        """Back-substitute, then linearise x[dst] speculatively (its cost becomes trial scalar 0)."""
        d = _f64(delta_cam)
        self._loss_forget(dst)
        self._chk(self.lib.mcba_step_linearize(self.handle, _p(d), float(lam), src, dst))

    def accept_linearization(self):
        self._chk(self.lib.mcba_accept_linearization(self.handle))

    def get_trial(self):
        self._chk(self.lib.mcba_get_trial(self.handle, _p(self._trial)))
        return self._trial.copy()

    # ---- device-resident LM iteration (one host synchronisation per iteration)
    def lm_set_state(self, cost, lam, nu, sel, curv_floor=0.0, curv_switch=0.0):
        """curv_floor: curvature model of the next linearisations (0 = the handle's, set_curvature_floor); curv_switch > 0: the device decision
        switches it (csrc/mcba_lm.h: Triggs after an accepted step that gained less than this fraction of the cost, IRLS after a rejection)."""
        st = np.zeros(LM_STATE)
        st[:4] = cost, lam, nu, sel
        st[25], st[26] = curv_floor, curv_switch
        self._chk(self.lib.mcba_lm_set_state(self.handle, _p(st)))

    def lm_iterate(self, delta_cam, pred_cam, dcn2, xcn2, lam_min, lam_max):
        """backsub -> linearise trial -> decide (GPU) -> Schur-reduce -> fetch.  Returns (system views, trial, state):
        views of buffers that the next call overwrites."""
        self._dc[:] = delta_cam
        rc = self.lib.mcba_lm_iterate(self.handle, self._dc_p, pred_cam, dcn2, xcn2, lam_min, lam_max, self._all_p)
        if rc:
            self._chk(rc)
        return self._red_views, self._trial, self._state

    def lm_trial(self, delta_cam):
        self._dc[:] = delta_cam
        self._chk(self.lib.mcba_lm_trial(self.handle, self._dc_p))

    def lm_decide_reduce(self, pred_cam, dcn2, xcn2, lam_min, lam_max, rank_slot=0):
        self._chk(self.lib.mcba_lm_decide_reduce(self.handle, pred_cam, dcn2, xcn2, lam_min, lam_max, int(rank_slot)))

    def lm_rebuild(self, rank_slot=0):
        self._chk(self.lib.mcba_lm_rebuild(self.handle, int(rank_slot)))

    def lm_fetch(self):
        self._chk(self.lib.mcba_lm_fetch(self.handle, self._all_p))
        return self._red_views, self._trial, self._state

    # ---- device-resident LM loop (reduced system solved on the GPU, no host synchronisation per iteration)
    AUTO_RING = 16

    def lm_auto_config(self, ftol, xtol, gtol, lam_min, lam_max, fixed_mask=None):
        m = None if fixed_mask is None else np.ascontiguousarray(fixed_mask, dtype=np.uint8).tobytes()
        self._chk(self.lib.mcba_lm_auto_config(self.handle, ftol, xtol, gtol, lam_min, lam_max, m))
        self._auto_state = np.zeros(LM_STATE)
        self._auto_state_p = _p(self._auto_state)

    def lm_set_decrease_floor(self, dec_floor):
        """Floor of Nielsen's damping factor on accepted steps (0 = the classical 1/3)."""
        self._chk(self.lib.mcba_lm_set_decrease_floor(self.handle, float(dec_floor)))

    def lm_auto_solve(self, seq, decide=0):
        rc = self.lib.mcba_lm_auto_solve(self.handle, seq, int(decide))
        if rc:
            self._chk(rc)

    def lm_auto_trial(self, decide):
        rc = self.lib.mcba_lm_auto_trial(self.handle, int(decide))
        if rc:
            self._chk(rc)

    def lm_auto_reduce(self, decide, rank_slot=0):
        rc = self.lib.mcba_lm_auto_reduce(self.handle, int(decide), int(rank_slot))
        if rc:
            self._chk(rc)

    def lm_auto_tick(self, seq, rank_slot=0):
        rc = self.lib.mcba_lm_auto_tick(self.handle, seq, int(rank_slot))
        if rc:
            self._chk(rc)

    def lm_run(self, x0, ftol, xtol, gtol, lam0, lam_min, lam_max, dec_floor, curv_floor, curv_switch, max_nfev, max_steps, depth, rank_slot=0, fixed_mask=None):
        """The device-resident LM loop from x0 (None: what slot 0 holds) to termination in ONE C-ABI crossing (include/mcba.h: mcba_lm_run).
        Returns (status, rows (k, LM_STATE) -- the state every retired tick posted, row 0 = the solve of the start point --, rows the
        loop proper consumed, iterations)."""
        opt = np.array([ftol, xtol, gtol, lam0, lam_min, lam_max, dec_floor, curv_floor, curv_switch, float(min(max_nfev, 2 ** 62)), -1.0 if max_steps is None else float(max_steps), depth, rank_slot], dtype=np.float64)
        summary = np.zeros(4)
        m = None if fixed_mask is None else np.ascontiguousarray(fixed_mask, dtype=np.uint8).tobytes()
        if x0 is not None:
            x0 = _f64(x0)
            if x0.shape != (self.nx,):
                raise ValueError(f"x must have {self.nx} entries")
        rc = self.lib.mcba_lm_run(self.handle, None if x0 is None else _p(x0), _p(opt), m, _p(summary))
        if rc == ERR_NONFINITE:
            raise ValueError("Residuals are not finite in the initial point.")
        self._chk(rc)
        rows = np.empty((int(summary[1]), LM_STATE))
        self._chk(self.lib.mcba_lm_history(self.handle, _p(rows), rows.shape[0]))
        self._auto_state = np.zeros(LM_STATE)
        self._auto_state_p = _p(self._auto_state)
        return int(summary[0]), rows, int(summary[2]), int(summary[3])

    def lm_result(self, slot, lazy_grad=False):
        """(x, gradient) of parameter slot `slot`, each 12C + 6F (include/mcba.h: mcba_lm_result).  lazy_grad: the gradient is returned as a
        DeviceArray that is downloaded when (if) it is read; otherwise one device-to-host copy brings both."""
        if lazy_grad:
            x = np.empty(self.nx)
            buf = _h()
            self._chk(self.lib.mcba_lm_result(self.handle, int(slot), _p(x), None, ctypes.byref(buf)))
            return x, DeviceArray(self.lib, buf, (self.nx,))
        out = np.empty((2, self.nx))
        self._chk(self.lib.mcba_lm_result(self.handle, int(slot), _p(out), _p(out[1]), None))
        return out[0], out[1]

    def lm_auto_wait(self, seq):
        """State (LM_STATE doubles, a buffer the next call overwrites) that tick `seq` posted."""
        rc = self.lib.mcba_lm_auto_wait(self.handle, seq, self._auto_state_p)
        if rc:
            self._chk(rc)
        return self._auto_state

    def cam_step(self):
        d = np.empty(self.n)
        self._chk(self.lib.mcba_get_cam_step(self.handle, _p(d)))
        return d

    # ---- direct RCCL on the library's own reduce buffer (frame-sharded runs)
    def comm_init_from_torch(self, group=None):
        """Create an RCCL communicator over the ranks of an initialised torch.distributed group: rank 0 draws the
        unique id, torch.distributed broadcasts it.  Every step is agreed on by ALL ranks before the next one (a rank that
        failed alone would leave the others waiting inside a collective), so the call either succeeds everywhere or raises
        everywhere -- and the caller may then fall back to torch collectives consistently."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            try:
                buf = ctypes.create_string_buffer(128)
                self._chk(self.lib.mcba_comm_unique_id(buf))
                box[0] = buf.raw
            except Exception as e:  # noqa: BLE001 -- reported to every rank through the broadcast below
                box[0] = "error: %s" % e
        dist.broadcast_object_list(box, src=0, group=group)
        if not isinstance(box[0], bytes):
            raise McbaError(-1, "rank 0 could not draw an RCCL unique id (%s)" % (box[0],))
        err = None
        try:
            self._chk(self.lib.mcba_comm_init(self.handle, box[0], rank, world))
        except Exception as e:  # noqa: BLE001
            err = e
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            if err is None:
                self.lib.mcba_comm_destroy(self.handle)
            raise McbaError(-1, "RCCL communicator could not be created on every rank (%s)" % (err or "another rank failed"))
        return rank, world

    def comm_count(self):
        """Ranks of the directly attached RCCL communicator as RCCL itself reports them (0: none attached)."""
        n = ctypes.c_int()
        self._chk(self.lib.mcba_comm_count(self.handle, ctypes.byref(n)))
        return n.value

    def comm_allreduce(self, offset, count):
        rc = self.lib.mcba_comm_allreduce(self.handle, offset, count)
        if rc:
            self._chk(rc)

    def frame_gradient(self):
        g = np.empty((self.F, 6))
        self._chk(self.lib.mcba_get_frame_gradient(self.handle, _p(g)))
        return g

    def enable_collective(self, device):
        """Own the reduce buffer as a torch tensor so torch.distributed (RCCL) can all-reduce it in place."""
        import torch

        self.reduce_tensor = torch.zeros(self.reduced_size(), dtype=torch.float64, device=device)
        self.bind_reduce_buffer(self.reduce_tensor.data_ptr())

    def synchronize(self):
        self._chk(self.lib.mcba_synchronize(self.handle))

    def profile_enable(self, on=True, only=None, stride=1, exact=False):
        """on=True: time every kernel; only=[names]: time just those; stride=k: bracket every k-th launch only
        (event records cost barrier packets on the stream: sample inside a measured region); exact: the fused k_gram kernel is timed by
        events on its dispatch -- its own begin / end, rocprofv3's figure (mcba_profile_exact)."""
        flag = int(bool(on))
        if on and only:
            names = self.lib.mcba_profile_names().decode().split("\n")
            flag = 0
            for k in only:
                flag |= 1 << (names.index(k) + 1)
        self._chk(self.lib.mcba_profile_enable(self.handle, flag))
        if on and stride > 1:
            self._chk(self.lib.mcba_profile_stride(self.handle, int(stride)))
        if on and exact:
            self._chk(self.lib.mcba_profile_exact(self.handle, 1))

    def profile_bracket_overhead(self, pairs=200):
        """Microseconds an empty HIP-event bracket reads on this handle's stream (include/mcba.h: mcba_profile_bracket_overhead)."""
        us = ctypes.c_double()
        self._chk(self.lib.mcba_profile_bracket_overhead(self.handle, int(pairs), ctypes.byref(us)))
        return us.value

    def profile_read(self):
        """{kernel name: (total ms, calls)} since the last read."""
        names = self.lib.mcba_profile_names().decode().split("\n")
        ms = (ctypes.c_double * len(names))()
        calls = (ctypes.c_int * len(names))()
        nk = ctypes.c_int()
        self._chk(self.lib.mcba_profile_read(self.handle, ms, calls, len(names), ctypes.byref(nk)))
        return {names[i]: (ms[i], calls[i]) for i in range(nk.value)}


def undistort_points(uvs, K4, dist5=None, iterations=5, device=0):
    """(…,2) pixel coordinates -> undistorted pixel coordinates (same camera matrix); NaN rows stay NaN."""
    lib = load_library()
    a = _f64(uvs)
    out = np.empty_like(a)
    k = _f64(K4)
    d = None if dist5 is None else _f64(dist5)
    rc = lib.mcba_undistort_points(a.size // 2, _p(a), _p(k), None if d is None else _p(d), int(iterations), int(device), _p(out))
    if rc != OK:
        raise McbaError(rc, lib.mcba_last_error().decode())
    return out


def pose_pairwise(poses, edges, device=0):
    """Median relative transforms of camera pairs over a caller's pose array (C,F,6) (include/mcba.h: mcba_pose_pairwise)."""
    lib = load_library()
    ps = _f64(poses)
    e = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
    out, cnt = np.empty((len(e), 6)), np.empty(len(e))
    rc = lib.mcba_pose_pairwise(ps.shape[0], ps.shape[1], _p(ps), e.ctypes.data_as(_ip), len(e), int(device), _p(out), _p(cnt))
    if rc != OK:
        raise McbaError(rc, lib.mcba_last_error().decode())
    return out, cnt


def pose_consensus(poses, extrinsics, device=0):
    """Consensus board poses (F,6) of a caller's pose array (C,F,6) under the extrinsics (C,6) (include/mcba.h: mcba_pose_consensus)."""
    lib = load_library()
    ps = _f64(poses)
    ext = _f64(extrinsics).reshape(ps.shape[0], 6)
    out = np.empty((ps.shape[1], 6))
    rc = lib.mcba_pose_consensus(ps.shape[0], ps.shape[1], _p(ps), _p(ext), int(device), _p(out))
    if rc != OK:
        raise McbaError(rc, lib.mcba_last_error().decode())
    return out


def calib_normal_equations(uvs, objpoints, intr9, poses, device=0):
    """Single-camera calibration with the five-coefficient model (include/mcba.h: mcba_calib_normal_equations): uvs (V,N,2), objpoints (N,3),
    intr9 = fx fy cx cy k1 k2 p1 p2 k3, poses (V,6) -> (H (V,15,15) symmetric Gauss-Newton blocks over [intr9 | pose6], g (V,15), cost (V,)),
    residuals observed - predicted."""
    lib = load_library()
    uvs, obj, k, ps = _f64(uvs), _f64(objpoints), _f64(intr9), _f64(poses)
    V, N = uvs.shape[:2]
    if uvs.shape != (V, N, 2) or obj.shape != (N, 3) or k.shape != (9,) or ps.shape != (V, 6):
        raise ValueError("uvs (V,N,2), objpoints (N,3), intr9 (9,), poses (V,6) required")
    out = np.empty((V, 136))
    rc = lib.mcba_calib_normal_equations(V, N, _p(uvs), _p(obj), _p(k), _p(ps), int(device), _p(out))
    if rc != OK:
        raise McbaError(rc, lib.mcba_last_error().decode())
    H = np.zeros((V, 15, 15))
    iu = np.triu_indices(15)
    H[:, iu[0], iu[1]] = out[:, :120]
    H[:, iu[1], iu[0]] = out[:, :120]
    return H, out[:, 120:135].copy(), out[:, 135].copy()
