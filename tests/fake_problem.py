"""Test double for `ops.Problem` built on the CPU oracle -- TEST INFRASTRUCTURE ONLY.

It implements the same interface the host-side LM driver (solver.py) talks to, so the driver's logic
(damping schedule, termination, fixed intrinsics, frame sharding + all-reduce layout) can be exercised
without a GPU.  The product never constructs it; `ops.Problem` (libmcba.so) is the only product backend."""
import numpy as np

from oracle import ba_oracle as orc


class OracleProblem:
    def __init__(self, uvs, objpoints, loss="soft_l1", f_scale=1.0, device=None, stream=None):
        self.uvs, self.obj = np.asarray(uvs, float), np.asarray(objpoints, float)
        self.x_scale = None
        self.C, self.F, self.N = self.uvs.shape[:3]
        self.n = 12 * self.C
        self.nx = self.n + 6 * self.F
        self.nsys = self.n * self.n + 3 * self.n + 16
        self.loss, self.f_scale = loss, f_scale
        self.x = [np.zeros(self.nx), np.zeros(self.nx)]
        self.reduce_tensor = None
        self._red = np.zeros(self.nsys + 8)
        self.calls = dict(linearize=0, build_reduced=0, step=0)
        self.curv_floor = orc.CURV_FLOOR

    def enable_collective(self, device="cpu"):
        import torch

        self.reduce_tensor = torch.zeros(self.nsys + 8, dtype=torch.float64)
        self._red = self.reduce_tensor.numpy()

    # ---- box constraints and the working set of the bounded loop (ops.Problem.set_bounds / set_frozen), in numpy
    def set_bounds(self, lo, hi):
        self.lo = None if lo is None else np.asarray(lo, float)
        self.hi = None if hi is None else np.asarray(hi, float)

    def set_frozen(self, mask):
        self.frozen_f = None if mask is None else np.asarray(mask, bool)[self.n:].reshape(self.F, 6)

    def _frozen_blocks(self):
        """(V, gf, W) with the frozen frame coordinates taken out: identity row / column in V_f, zero gradient entry, zero column of W."""
        U, gc, V, gf, W, cost = self.lin
        fr = getattr(self, "frozen_f", None)
        if fr is None or not fr.any():
            return V, gf, W
        V, gf, W = V.copy(), gf.copy(), W.copy()
        for f, k in zip(*np.nonzero(fr)):
            V[f, k, :] = 0.0
            V[f, :, k] = 0.0
            V[f, k, k] = 1.0
            gf[f, k] = 0.0
            W[:, f, :, k] = 0.0
        return V, gf, W

    def set_curvature_floor(self, floor):
        old, self.curv_floor = self.curv_floor, float(floor)
        return old

    def set_params(self, slot, x):
        self.x[slot] = np.array(x, dtype=float)

    def get_params(self, slot):
        return self.x[slot].copy()

    def linearize(self, slot):
        self.calls["linearize"] += 1
        self.lin = orc.normal_equations(self.x[slot], self.uvs, self.obj, self.loss, self.f_scale, self.curv_floor)
        self.lin_x = self.x[slot].copy()

    def build_reduced(self, lam, rank_slot=0):
        self.calls["build_reduced"] += 1
        U, gc, V, gf, W, cost = self.lin
        n = self.n
        Df2 = np.stack([np.where(np.diag(V[f]) > 0, np.diag(V[f]), 1.0) for f in range(self.F)])
        if self.x_scale is not None:  # numeric x_scale: fixed D = 1 / x_scale^2
            Df2 = 1.0 / self.x_scale[self.n:].reshape(self.F, 6) ** 2
        Vm, gfm, Wm = self._frozen_blocks()
        fr = getattr(self, "frozen_f", None)
        if fr is not None and fr.any():
            Df2 = np.where(fr, 0.0, Df2)   # (the frozen coordinates' diagonal entries are the identity's, undamped)
        S, rhs = orc.schur_reduce(U, gc, Vm, gfm, Wm, lam, np.zeros((self.C, 12)), Df2)
        r = self._red
        r[:] = 0
        r[: n * n] = S.ravel()
        r[n * n : n * n + n] = rhs
        r[n * n + n : n * n + 2 * n] = np.concatenate([np.diag(U[c]) for c in range(self.C)])
        r[n * n + 2 * n : n * n + 3 * n] = gc.ravel()
        sc = r[n * n + 3 * n : n * n + 3 * n + 16]
        sc[0] = cost
        sc[4 + rank_slot] = np.abs(gfm).max()
        self.Df2, self.lam = Df2, lam

    def get_reduced(self):
        n, r = self.n, self._red
        return dict(S0=r[: n * n].reshape(n, n).copy(), rhs=r[n * n : n * n + n].copy(), diagU=r[n * n + n : n * n + 2 * n].copy(),
                    gc=r[n * n + 2 * n : n * n + 3 * n].copy(), scal=r[n * n + 3 * n : n * n + 3 * n + 16].copy())

    def step(self, dc, lam, src, dst):
        self.calls["step"] += 1
        U, gc, V, gf, W, cost = self.lin
        Vm, gfm, Wm = self._frozen_blocks()
        df = orc.back_substitute(np.asarray(dc), Vm, gfm, Wm, lam, self.Df2)
        xs = self.x[src]
        self.x[dst] = xs + np.concatenate([dc, df.ravel()])
        if getattr(self, "lo", None) is not None:   # the trial point, projected onto the box (ops: k_clip)
            self.x[dst] = np.minimum(np.maximum(self.x[dst], self.lo), self.hi)
        f = orc.residuals(self.x[dst], self.uvs, self.obj)
        t = self._red[self.nsys :]
        t[:] = 0
        t[0] = orc.robust_cost(f, self.loss, self.f_scale)
        t[1] = np.sum(df * (lam * self.Df2 * df - gfm))
        t[2] = np.sum(df * df)
        t[3] = np.sum(xs[self.n :] ** 2)
        t[4] = f.size

    def step_linearize(self, dc, lam, src, dst):
        self.step(dc, lam, src, dst)
        self.calls["linearize"] += 1
        self.lin_trial = orc.normal_equations(self.x[dst], self.uvs, self.obj, self.loss, self.f_scale, self.curv_floor)

    def accept_linearization(self):
        self.lin = self.lin_trial

    def residuals(self, slot):
        r = self.uvs - orc.predict_from_x(self.x[slot], self.C, self.obj)
        return np.where(np.isnan(self.uvs), 0.0, r)

    def close(self):
        pass

    def set_x_scale(self, x_scale):
        self.x_scale = None if x_scale is None else np.asarray(x_scale, float)

    def set_loss(self, loss, f_scale=1.0):
        self.loss, self.f_scale = loss, f_scale

    @property
    def loss_is_callable(self):
        return callable(self.loss)

    # ---- the pre-filter interface of ops.Problem (api.select_frames), in numpy
    def _errors(self, slot):
        r = self.uvs - orc.predict_from_x(self.x[slot], self.C, self.obj)
        return np.sqrt((r ** 2).sum(-1))  # NaN wherever either coordinate is missing, like np.linalg.norm(obs - pred)

    def frame_errors(self, slot):
        import warnings

        self._err = self._errors(slot)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            mean = np.nanmean(self._err, axis=-1)
        full = (~np.isnan(self.uvs).any(-1)).sum(-1).astype(float)
        return mean, full

    def error_median(self, frame_mask=None):
        e = self._err if frame_mask is None else self._err[:, np.asarray(frame_mask, bool)]
        v = e[~np.isnan(e)]
        return (float(np.median(v)) if v.size else float("nan")), int(v.size)

    def error_histogram(self, frame_mask, prefix, pass_):
        if frame_mask is not None:
            self._hmask = np.asarray(frame_mask, bool)
        e = self._err[:, self._hmask]
        keys = e[~np.isnan(e)].view(np.uint64)
        if pass_ > 0:
            keys = keys[(keys >> np.uint64(64 - 8 * pass_)) == np.uint64(prefix)]
        return np.bincount(((keys >> np.uint64(56 - 8 * pass_)) & np.uint64(255)).astype(np.int64), minlength=256).astype(np.uint64)

    def subset(self, frames, loss=None, f_scale=None):
        return OracleProblem(self.uvs[:, np.asarray(frames, int)], self.obj, loss or self.loss, self.f_scale if f_scale is None else f_scale)

    def residuals_detach(self, slot):
        r = self.residuals(slot)

        class _Host:
            def download(self_inner):
                return r

        return _Host()

    def get_trial(self):
        return self._red[self.nsys :].copy()

    def frame_gradient(self):
        return self.lin[3].copy()
