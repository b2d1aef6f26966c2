"""CPU oracle for the bundle-adjustment hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain-numpy restatement of the reference's algorithm for
`multicam_calibration.bundle_adjust` and the geometry it bottoms out in.  It is
the *checker* for the HIP path: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  Nothing under
`multicam-calibration_amd/` imports it, and the product path never falls back to
it (the product raises if `libmcba.so` is missing).

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so the pins are outputs of the reference itself, generated in
the build container by `tests/golden/make_golden.py` (which imports the
reference's `bundle_adjustment.py`/`geometry.py` unmodified) and committed as
`tests/golden/*.npz`.  `tests/test_oracle_golden.py` checks every function here
against those files.

Third-party arithmetic on this path: `scipy.optimize.least_squares`
(scipy, unpinned by the reference -- setup.cfg:14-15; 1.15.3 in this image),
called exactly as the reference calls it (bundle_adjustment.py:301-313).

Each function cites the reference lines it follows.  Parameter layout (a7 in
SURVEY.md section 8): x = [C x (fx fy cx cy k1 k2 rx ry rz tx ty tz) | F x (rx ry rz tx ty tz)].
"""
import warnings

import numpy as np
import scipy.sparse as sp
from scipy.optimize import least_squares

EPS = np.finfo(float).eps


# --------------------------------------------------------------------------- geometry
def skew(v):
    """[v]x for (...,3) vectors."""
    v = np.asarray(v, dtype=float)
    S = np.zeros(v.shape[:-1] + (3, 3))
    S[..., 0, 1] = -v[..., 2]
    S[..., 0, 2] = v[..., 1]
    S[..., 1, 0] = v[..., 2]
    S[..., 1, 2] = -v[..., 0]
    S[..., 2, 0] = -v[..., 1]
    S[..., 2, 1] = v[..., 0]
    return S


def rodrigues(r):
    """Rotation vector -> matrix, reference convention (geometry.py:8-35):
    axis = r/theta with the division skipped at theta == 0, so R(0) = I exactly."""
    r = np.asarray(r, dtype=float)
    theta = np.linalg.norm(r, axis=-1)[..., None, None]
    A = skew(r) / np.where(theta == 0, 1.0, theta)
    return np.eye(3) + np.sin(theta) * A + (1.0 - np.cos(theta)) * (A @ A)


def rodrigues_inv(R):
    """Rotation matrix -> vector (geometry.py:38-65).  Used only by the gauge
    alignment / synthetic-data helpers, not on the hot path."""
    R = np.asarray(R, dtype=float)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], axis=-1)
    c = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) / 2.0, -1.0, 1.0)
    theta = np.arccos(c)[..., None]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    n = n + (n == 0)
    return v * theta / n


def rigid_apply(t6, pts):
    """X -> R(t6[:3]) X + t6[3:]   (geometry.py:128-175; the reference goes through
    4x4 homogeneous matrices, the closed form differs by round-off only)."""
    R = rodrigues(t6[..., :3])
    return np.einsum("...ij,...j->...i", R, pts) + t6[..., 3:]


def embed_calib_objpoints(calib_objpoints, calib_poses):
    """Board points -> world, per frame (bundle_adjustment.py:10-30): (F,N,3)."""
    R = rodrigues(calib_poses[:, :3])  # (F,3,3)
    return np.einsum("fij,nj->fni", R, calib_objpoints) + calib_poses[:, None, 3:]


def project_points(points, extrinsics, fx, fy, cx, cy, k1, k2):
    """Pinhole + 2-term radial projection (geometry.py:277-325).
    a=x/z, b=y/z, s=a^2+b^2, d=1+k1 s+k2 s^2, u=fx a d+cx, v=fy b d+cy."""
    Xc = np.einsum("ij,...j->...i", rodrigues(extrinsics[:3]), points) + extrinsics[3:]
    a = Xc[..., 0] / Xc[..., 2]
    b = Xc[..., 1] / Xc[..., 2]
    s = a * a + b * b
    d = 1.0 + k1 * s + k2 * s * s
    return np.stack([fx * a * d + cx, fy * b * d + cy], axis=-1)


# --------------------------------------------------------------------------- parameter vector
def serialize_params(all_extrinsics, all_intrinsics, calib_poses):
    """bundle_adjustment.py:128-157.  p1, p2, k3 of dist_coefs are dropped."""
    cams = []
    for ext, (K, dist) in zip(all_extrinsics, all_intrinsics):
        cams.append([K[0, 0], K[1, 1], K[0, 2], K[1, 2], dist[0], dist[1], *ext])
    return np.concatenate([np.asarray(cams, dtype=float).ravel(), np.asarray(calib_poses, dtype=float).ravel()])


def deserialize_params(x, n_cameras):
    """bundle_adjustment.py:160-192.  dist_coefs come back as (k1,k2,0,0,0)."""
    cam = np.asarray(x[: 12 * n_cameras]).reshape(n_cameras, 12)
    all_extrinsics = cam[:, 6:].copy()
    all_intrinsics = []
    for c in range(n_cameras):
        K = np.eye(3)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = cam[c, :4]
        all_intrinsics.append((K, np.array([cam[c, 4], cam[c, 5], 0.0, 0.0, 0.0])))
    return all_extrinsics, all_intrinsics, np.asarray(x[12 * n_cameras :]).reshape(-1, 6).copy()


def predict_from_x(x, n_cameras, calib_objpoints):
    """(C,F,N,2) predicted detections from the flat vector (bundle_adjustment.py:33-63)."""
    cam = np.asarray(x[: 12 * n_cameras]).reshape(n_cameras, 12)
    poses = np.asarray(x[12 * n_cameras :]).reshape(-1, 6)
    Xw = embed_calib_objpoints(calib_objpoints, poses)
    return np.stack([project_points(Xw, cam[c, 6:], *cam[c, :6]) for c in range(n_cameras)])


def predict_calib_uvs(all_extrinsics, all_intrinsics, calib_objpoints, calib_poses):
    return predict_from_x(serialize_params(all_extrinsics, all_intrinsics, calib_poses), len(all_extrinsics), calib_objpoints)


def residuals(x, all_calib_uvs, calib_objpoints):
    """observed - predicted, NaN scalars removed one by one, C-order over
    (cam, frame, point, uv)  (bundle_adjustment.py:66-98)."""
    pred = predict_from_x(x, all_calib_uvs.shape[0], calib_objpoints)
    return (all_calib_uvs - pred)[~np.isnan(all_calib_uvs)]


def sparsity_csr(all_calib_uvs):
    """Same pattern as bundle_adjustment_sparsity (bundle_adjustment.py:101-125):
    18 ones per row -- columns 12c..12c+11 and 12C+6f..12C+6f+5 -- built directly as
    CSR instead of the reference's LIL (identical indices, see the golden test)."""
    C, F, N, _ = all_calib_uvs.shape
    mask = ~np.isnan(all_calib_uvs)
    cam_ix = np.broadcast_to(np.arange(C)[:, None, None, None], mask.shape)[mask]
    frm_ix = np.broadcast_to(np.arange(F)[None, :, None, None], mask.shape)[mask]
    m = cam_ix.size
    cols = np.concatenate([cam_ix[:, None] * 12 + np.arange(12), 12 * C + frm_ix[:, None] * 6 + np.arange(6)], axis=1)
    indptr = np.arange(m + 1, dtype=np.int64) * 18
    return sp.csr_matrix((np.ones(m * 18, dtype=int), cols.ravel(), indptr), shape=(m, 12 * C + 6 * F))


# --------------------------------------------------------------------------- analytic Jacobian
def right_jacobian(r):
    """Jr(r) = I - b [r]x + c [r]x^2, b = (1-cos t)/t^2, c = (t-sin t)/t^3  (R(r+e) ~ R(r) Exp(Jr e)).
    Equal to (r r^T + (R^T - I)[r]x)/|r|^2 of SURVEY.md section 8a; that closed form cancels
    catastrophically for tiny non-zero |r|, so a Taylor series is used below |r|^2 = 1e-6."""
    r = np.asarray(r, dtype=float)
    th2 = np.sum(r * r, axis=-1)[..., None, None]
    small = th2 < 1e-6
    th2s = np.where(small, 1.0, th2)
    th = np.sqrt(th2s)
    b = np.where(small, 0.5 - th2 / 24 + th2**2 / 720, (1 - np.cos(th)) / th2s)
    c = np.where(small, 1 / 6 - th2 / 120 + th2**2 / 5040, (th - np.sin(th)) / (th * th2s))
    K = skew(r)
    return np.eye(3) - b * K + c * (K @ K)


def drot_point(r, X):
    """G(r, X) = d(R(r) X)/dr = -R [X]x Jr(r), (...,3,3); -[X]x at r = 0."""
    return -(rodrigues(r) @ skew(X)) @ right_jacobian(r)


def jacobian_blocks(x, n_cameras, calib_objpoints):
    """Per point-observation Jacobian blocks of the PREDICTION (not the residual):
    Jc (C,F,N,2,12) w.r.t. the camera's 12 params, Jf (C,F,N,2,6) w.r.t. the frame pose."""
    C = n_cameras
    cam = np.asarray(x[: 12 * C]).reshape(C, 12)
    poses = np.asarray(x[12 * C :]).reshape(-1, 6)
    F, N = poses.shape[0], calib_objpoints.shape[0]
    Rf = rodrigues(poses[:, :3])
    Xw = np.einsum("fij,nj->fni", Rf, calib_objpoints) + poses[:, None, 3:]
    Gf = drot_point(poses[:, None, :3], calib_objpoints[None, :, :])  # (F,N,3,3)
    Jc = np.zeros((C, F, N, 2, 12))
    Jf = np.zeros((C, F, N, 2, 6))
    for c in range(C):
        fx, fy, cx, cy, k1, k2 = cam[c, :6]
        rho, t = cam[c, 6:9], cam[c, 9:12]
        Rc = rodrigues(rho)
        Xc = np.einsum("ij,fnj->fni", Rc, Xw) + t
        iz = 1.0 / Xc[..., 2]
        a, b = Xc[..., 0] * iz, Xc[..., 1] * iz
        s = a * a + b * b
        d = 1 + k1 * s + k2 * s * s
        dp = k1 + 2 * k2 * s
        Dab = np.empty((F, N, 2, 2))
        Dab[..., 0, 0] = fx * (d + 2 * a * a * dp)
        Dab[..., 0, 1] = 2 * fx * a * b * dp
        Dab[..., 1, 0] = 2 * fy * a * b * dp
        Dab[..., 1, 1] = fy * (d + 2 * b * b * dp)
        Pz = np.zeros((F, N, 2, 3))
        Pz[..., 0, 0] = iz
        Pz[..., 0, 2] = -a * iz
        Pz[..., 1, 1] = iz
        Pz[..., 1, 2] = -b * iz
        DX = Dab @ Pz
        Jc[c, ..., 0, 0] = a * d
        Jc[c, ..., 1, 1] = b * d
        Jc[c, ..., 0, 2] = 1
        Jc[c, ..., 1, 3] = 1
        Jc[c, ..., 0, 4] = fx * a * s
        Jc[c, ..., 1, 4] = fy * b * s
        Jc[c, ..., 0, 5] = fx * a * s * s
        Jc[c, ..., 1, 5] = fy * b * s * s
        Jc[c, ..., 6:9] = DX @ drot_point(rho, Xw)
        Jc[c, ..., 9:12] = DX
        DXR = DX @ Rc
        Jf[c, ..., 0:3] = DXR @ Gf
        Jf[c, ..., 3:6] = DXR
    return Jc, Jf


def jacobian_csr(x, all_calib_uvs, calib_objpoints):
    """Analytic Jacobian of residuals() as CSR in the reference's row/column order."""
    C, F, N, _ = all_calib_uvs.shape
    Jc, Jf = jacobian_blocks(x, C, calib_objpoints)
    mask = ~np.isnan(all_calib_uvs)
    data = -np.concatenate([Jc, Jf], axis=-1)[mask]  # residual = obs - pred
    pat = sparsity_csr(all_calib_uvs)
    return sp.csr_matrix((data.ravel(), pat.indices, pat.indptr), shape=pat.shape)


# --------------------------------------------------------------------------- robust loss (scipy/optimize/_lsq/least_squares.py:160-227, common.py:720-731)
def loss_rho(z, loss):
    """rho(z), rho'(z), rho''(z) for scipy's built-in losses, z = (f/f_scale)^2 -- or from least_squares' CALLABLE form, a function
    z -> array (3, m) (least_squares.py:160-227; construct_loss_function applies the f_scale factors exactly as for the names)."""
    z = np.asarray(z, dtype=float)
    if callable(loss):
        rho = np.asarray(loss(z), dtype=float)
        if rho.shape != (3,) + z.shape:
            raise ValueError("The return value of `loss` callable has wrong shape.")
        return rho[0], rho[1], rho[2]
    if loss == "linear":
        return z.copy(), np.ones_like(z), np.zeros_like(z)
    if loss == "soft_l1":
        t = 1 + z
        return 2 * (t**0.5 - 1), t**-0.5, -0.5 * t**-1.5
    if loss == "huber":
        m = z <= 1
        with np.errstate(divide="ignore", invalid="ignore"):
            r0 = np.where(m, z, 2 * z**0.5 - 1)
            r1 = np.where(m, 1.0, z**-0.5)
            r2 = np.where(m, 0.0, -0.5 * z**-1.5)
        return r0, r1, r2
    if loss == "cauchy":
        return np.log1p(z), 1 / (1 + z), -1 / (1 + z) ** 2
    if loss == "arctan":
        t = 1 + z * z
        return np.arctan(z), 1 / t, -2 * z / t**2
    raise ValueError(loss)


def robust_cost(f, loss="soft_l1", f_scale=1.0):
    """0.5 * f_scale^2 * sum rho((f/f_scale)^2)  (least_squares.py:212-227 with cost_only)."""
    return 0.5 * f_scale**2 * np.sum(loss_rho((f / f_scale) ** 2, loss)[0])


def robust_scales(f, loss="soft_l1", f_scale=1.0):
    """Row scale for J and the rescaled residual (common.py:720-731).
    Returns (J_scale, f_scaled) with J_scale = sqrt(max(rho' + 2 rho'' f^2, EPS))."""
    z = (f / f_scale) ** 2
    _, r1, r2 = loss_rho(z, loss)
    r2 = r2 / f_scale**2
    js = r1 + 2 * r2 * f * f
    js = np.sqrt(np.where(js < EPS, EPS, js))
    return js, f * r1 / js


# --------------------------------------------------------------------------- wrapper logic (bundle_adjustment.py:265-296)
def prefilter_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames=10000, outlier_threshold=None):
    """Frame selection of bundle_adjust: complete in >= 2 cameras, worst-camera mean error
    below 5x the nan-median (or the given threshold), then a global-RNG subsample.
    Returns (use_frames, threshold, n_excluded, printed_line)."""
    use = np.nonzero((~np.isnan(all_calib_uvs).any((-1, -2))).sum(0) > 1)[0]
    pred = predict_calib_uvs(all_extrinsics, all_intrinsics, calib_objpoints, calib_poses[use])
    err = np.linalg.norm(all_calib_uvs[:, use] - pred, axis=-1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        worst = np.nanmax(np.nanmean(err, axis=-1), axis=0)
    if outlier_threshold is None:
        outlier_threshold = 5 * np.nanmedian(err)
    exclude = np.nan_to_num(worst) > outlier_threshold
    use = use[~exclude]
    line = f"Excluding {int(exclude.sum())} out of {len(use)} frames based on an outlier threshold of {outlier_threshold}"
    if not (n_frames is None or n_frames > len(use)):
        use = np.random.choice(use, n_frames, replace=False)
    return use, outlier_threshold, int(exclude.sum()), line


def bundle_adjust(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames=10000, outlier_threshold=None, **opt_kwargs):
    """The reference's CPU path (bundle_adjustment.py:195-327): same pre-filter, same
    x0, same third-party least_squares call with the same defaults."""
    C = all_calib_uvs.shape[0]
    use, _, _, line = prefilter_frames(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, n_frames, outlier_threshold)
    print(line)
    uvs = all_calib_uvs[:, use]
    A = sparsity_csr(uvs)
    x0 = serialize_params(all_extrinsics, all_intrinsics, calib_poses[use])
    kw = dict(verbose=2, x_scale="jac", ftol=1e-4, method="trf", loss="soft_l1")
    kw.update(opt_kwargs)
    result = least_squares(residuals, x0, jac_sparsity=A, **kw, args=(uvs, calib_objpoints))
    ext, intr, poses = deserialize_params(result.x, C)
    return ext, intr, poses, use, result


# --------------------------------------------------------------------------- dense normal equations / Schur (checker for the GPU assembly)
CURV_FLOOR = 1.0  # csrc/mcba_math.h: the LM's curvature weight is max(Triggs, floor * rho'); 1.0 = the IRLS weight rho' = the library's default (mcba_set_curvature_floor), 0.1 = Triggs with a floor


def normal_equations(x, all_calib_uvs, calib_objpoints, loss="soft_l1", f_scale=1.0, curv_floor=CURV_FLOOR):
    """Dense pieces of the robust Gauss-Newton system at x, for SMALL problems:
    U (C,12,12), gc (C,12), V (F,6,6), gf (F,6), W (C,F,12,6), cost.
    The gradient is J^T (rho' f) exactly as scipy forms it (J and f rescaled, common.py:720-731); the
    curvature weight is scipy's rho' + 2 rho'' f^2 floored at curv_floor * rho' (0 reproduces scipy's J~^T J~)."""
    C, F, N, _ = all_calib_uvs.shape
    Jc, Jf = jacobian_blocks(x, C, calib_objpoints)
    pred = predict_from_x(x, C, calib_objpoints)
    valid = ~np.isnan(all_calib_uvs)
    f = np.where(valid, all_calib_uvs - pred, 0.0)
    js, fs = robust_scales(f, loss, f_scale)
    rho1 = loss_rho((f / f_scale) ** 2, loss)[1]
    gw = np.where(valid, rho1 * f, 0.0)                       # = js * fs
    w = np.where(valid, np.maximum(js * js, curv_floor * rho1), 0.0)
    Jc, Jf = -Jc, -Jf
    U = np.einsum("cfnri,cfnr,cfnrj->cij", Jc, w, Jc)
    gc = np.einsum("cfnri,cfnr->ci", Jc, gw)
    V = np.einsum("cfnri,cfnr,cfnrj->fij", Jf, w, Jf)
    gf = np.einsum("cfnri,cfnr->fi", Jf, gw)
    W = np.einsum("cfnri,cfnr,cfnrj->cfij", Jc, w, Jf)
    cost = robust_cost(f[valid], loss, f_scale)
    return U, gc, V, gf, W, cost


def schur_reduce(U, gc, V, gf, W, lam, Dc2, Df2):
    """Reduced camera system of (J^T J + lam D^2) delta = -g:
    S = U + lam Dc2 - sum_f W_f (V_f + lam Df2_f)^-1 W_f^T,  rhs = -gc + sum_f W_f (V_f + lam Df2_f)^-1 gf_f."""
    C, F = W.shape[:2]
    S = np.zeros((12 * C, 12 * C))
    for c in range(C):
        S[12 * c : 12 * c + 12, 12 * c : 12 * c + 12] = U[c]
    S += lam * np.diag(Dc2.ravel())
    rhs = -gc.ravel().copy()
    for f in range(F):
        Wf = W[:, f].reshape(12 * C, 6)
        Vi = np.linalg.inv(V[f] + lam * np.diag(Df2[f]))
        S -= Wf @ Vi @ Wf.T
        rhs += Wf @ Vi @ gf[f]
    return S, rhs


def back_substitute(dc, V, gf, W, lam, Df2):
    """delta_f = -(V_f + lam Df2_f)^-1 (gf_f + W_f^T delta_c)."""
    C, F = W.shape[:2]
    out = np.zeros((F, 6))
    for f in range(F):
        Wf = W[:, f].reshape(12 * C, 6)
        out[f] = -np.linalg.solve(V[f] + lam * np.diag(Df2[f]), gf[f] + Wf.T @ dc)
    return out


# --------------------------------------------------------------------------- gauge alignment (hard part 1)
def to_matrix(t6):
    T = np.zeros(t6.shape[:-1] + (4, 4))
    T[..., :3, :3] = rodrigues(t6[..., :3])
    T[..., :3, 3] = t6[..., 3:]
    T[..., 3, 3] = 1
    return T


def to_vector(T):
    return np.concatenate([rodrigues_inv(T[..., :3, :3]), T[..., :3, 3]], axis=-1)


def gauge_align(all_extrinsics, calib_poses, target_ext0):
    """Move the world frame so that camera 0's extrinsic equals `target_ext0`.
    Cameras: T_c <- T_c G^-1, boards: P_f <- G P_f with G = target^-1 T_0 ... chosen so
    every camera-from-board transform T_c P_f (hence every prediction) is unchanged."""
    Tc = to_matrix(np.asarray(all_extrinsics))
    Pf = to_matrix(np.asarray(calib_poses))
    G = np.linalg.inv(to_matrix(np.asarray(target_ext0))) @ Tc[0]
    return to_vector(Tc @ np.linalg.inv(G)), to_vector(G @ Pf)


def invariants(all_extrinsics, calib_poses):
    """Gauge-invariant description: camera-0-from-camera-c transforms and camera-0-from-board transforms."""
    Tc = to_matrix(np.asarray(all_extrinsics))
    Pf = to_matrix(np.asarray(calib_poses))
    return Tc @ np.linalg.inv(Tc[0]), Tc[0] @ Pf
