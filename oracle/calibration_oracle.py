"""CPU restatement of the single-camera model behind the reference's get_intrinsics() / estimate_pose() when the five-coefficient distortion
model is asked for (multicam_calibration/calibration.py:11-71: cv2.calibrateCamera with flags = CALIB_FIX_K3 * fix_k3 + CALIB_ZERO_TANGENT_DIST *
zero_tangent_dist; :74-113: cv2.solvePnP with the coefficients it returned).

TEST INFRASTRUCTURE ONLY: imported by tests/ -- never by the product (tests/test_abi.py enforces it).

OpenCV is a third-party dependency of the reference that this image does not have (import cv2 fails; no fixtures of its outputs exist in the
reference), so the restatement follows OpenCV's PUBLISHED camera model (calib3d documentation, "Camera Calibration and 3D Reconstruction"):
    x = X_c / Z_c, y = Y_c / Z_c, r2 = x^2 + y^2
    x'' = x (1 + k1 r2 + k2 r2^2 + k3 r2^3) + 2 p1 x y + p2 (r2 + 2 x^2)
    y'' = y (1 + k1 r2 + k2 r2^2 + k3 r2^3) + p1 (r2 + 2 y^2) + 2 p2 x y
    u = fx x'' + cx, v = fy y'' + cy                      distortion vector order: (k1, k2, p1, p2, k3)
and calibrateCamera's objective (the sum of squared reprojection errors over all views, minimised over the intrinsics not held by a flag and
every view's pose).  PARITY UNPINNED against cv2 numbers: the GPU path is compared with this restatement (finite-difference derivatives, scipy's
least_squares as the minimiser) and with the synthetic truth."""
import numpy as np
from scipy.optimize import least_squares
from scipy.optimize._numdiff import approx_derivative


def rodrigues(r):
    """geometry.py:8-35 (theta = 0: divide by 1)."""
    r = np.asarray(r, dtype=float)
    theta = np.linalg.norm(r)
    A = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]]) / (theta if theta != 0 else 1.0)
    return np.eye(3) + np.sin(theta) * A + (1 - np.cos(theta)) * (A @ A)


def project5(obj, pose, intr9):
    """(N,3) board points, pose (6,) board -> camera, intr9 = fx fy cx cy k1 k2 p1 p2 k3 -> (N,2) pixels."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = intr9
    Xc = np.asarray(obj, dtype=float) @ rodrigues(pose[:3]).T + np.asarray(pose[3:], dtype=float)
    x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2**2 + k3 * r2**3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], axis=-1)


def view_residuals(params15, uv, obj):
    """observed - predicted of one view, (2N,) in (point, uv) order; params15 = intr9 | pose6."""
    return (np.asarray(uv, dtype=float) - project5(obj, params15[9:], params15[:9])).ravel()


def view_normal_equations(uv, obj, intr9, pose):
    """(H (15,15), g (15,), cost) of one view from a 3-point finite-difference Jacobian of view_residuals."""
    p = np.concatenate([intr9, pose])
    J = approx_derivative(view_residuals, p, method="3-point", args=(uv, obj))
    r = view_residuals(p, uv, obj)
    return J.T @ J, J.T @ r, 0.5 * float(r @ r)


def refine(uvs, obj, intr9, poses, free9):
    """calibrateCamera's objective minimised with scipy's least_squares over the free intrinsics and every pose, from the given start."""
    uvs, intr9, poses = np.asarray(uvs, dtype=float), np.asarray(intr9, dtype=float), np.asarray(poses, dtype=float)
    free9 = np.asarray(free9, dtype=bool)
    V = len(uvs)

    def unpack(z):
        k = intr9.copy()
        k[free9] = z[: free9.sum()]
        return k, z[free9.sum():].reshape(V, 6)

    def fun(z):
        k, ps = unpack(z)
        return np.concatenate([(uvs[v] - project5(obj, ps[v], k)).ravel() for v in range(V)])

    z0 = np.concatenate([intr9[free9], poses.ravel()])
    res = least_squares(fun, z0, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-12, max_nfev=200)
    k, ps = unpack(res.x)
    return k, ps, res.cost


# ---------------------------------------------------------------------------------------------------------------------------------
# calibrate()'s per-view closed-form starts and its pose graph, in numpy (round 6: the product runs them on the GPU -- csrc/mcba_pnp.hip --
# and these are what its kernels are checked against).
#   pose graph: restates multicam_calibration/calibration.py:116-143 (estimate_pairwise_camera_transform) and :239-277
#   (consensus_calib_poses) with geometry.py:38-56,155-196 (rodrigues_inv, the 6-vector <-> 4x4 maps); PINNED to the reference's own
#   outputs by tests/golden/calibration_graph.npz (tests/test_calibration_cpu.py).
#   homographies / poses_from_homographies / undistort_normalized: the steps OpenCV takes inside cv2.calibrateCamera / cv2.solvePnP for a
#   planar board (Hartley-normalised DLT -- Zhang 2000 appendix A --, pose from H = K [r1 r2 t], undistortPoints' fixed point); cv2 is absent:
#   parity with its numbers is unpinned, the tests check exact recovery on noise-free synthetic views.
na = np.newaxis


def rodrigues_batch(r):
    """geometry.py:8-35 for an array of rotation vectors (..., 3) -> (..., 3, 3)."""
    r = np.asarray(r, dtype=np.float64)
    theta = np.linalg.norm(r, axis=-1)[..., na, na]
    safe = np.where(theta == 0, 1.0, theta)
    A = np.zeros(r.shape[:-1] + (3, 3))
    A[..., 0, 1], A[..., 0, 2] = -r[..., 2], r[..., 1]
    A[..., 1, 0], A[..., 1, 2] = r[..., 2], -r[..., 0]
    A[..., 2, 0], A[..., 2, 1] = -r[..., 1], r[..., 0]
    A = A / safe
    return np.eye(3) + np.sin(theta) * A + (1 - np.cos(theta)) * (A @ A)


def rodrigues_inv(R):
    R = np.asarray(R, dtype=np.float64)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], axis=-1)
    theta = np.arccos((np.trace(R, axis1=-2, axis2=-1) - 1) / 2)[..., na]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    n = n + (n == 0)
    return v * theta / n


def get_transformation_matrix(t):
    t = np.asarray(t, dtype=np.float64)
    T = np.zeros(t.shape[:-1] + (4, 4))
    T[..., :3, :3] = rodrigues_batch(t[..., :3])
    T[..., :3, 3] = t[..., 3:]
    T[..., 3, 3] = 1
    return T


def get_transformation_vector(T):
    return np.concatenate([rodrigues_inv(T[..., :3, :3]), T[..., :3, 3]], axis=-1)


def estimate_pairwise_camera_transform(camera1_poses, camera2_poses):
    """Median over the common frames of T2 T1^-1, component-wise on the 6-vectors (calibration.py:116-143)."""
    camera1_poses, camera2_poses = np.asarray(camera1_poses), np.asarray(camera2_poses)
    common = ~np.isnan([camera1_poses, camera2_poses]).any((0, 2))
    T1 = get_transformation_matrix(camera1_poses[common])
    T2 = get_transformation_matrix(camera2_poses[common])
    return np.median(get_transformation_vector(T2 @ np.linalg.inv(T1)), axis=0)


def consensus_calib_poses(all_calib_poses, all_extrinsics):
    """Per-camera board poses mapped to world coordinates, nan-median over cameras (calibration.py:239-277)."""
    import warnings

    all_calib_poses = np.asarray(all_calib_poses, dtype=np.float64)
    world = np.full_like(all_calib_poses, np.nan)
    for i, (poses, transform) in enumerate(zip(all_calib_poses, all_extrinsics)):
        det = ~np.isnan(poses).any(axis=-1)
        T = np.linalg.inv(get_transformation_matrix(transform)) @ get_transformation_matrix(poses[det])
        world[i, det] = get_transformation_vector(T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        return np.nanmedian(world, axis=0)


def homographies(XY, uv):
    """Normalised DLT, batched: XY (N,2) board coordinates, uv (F,N,2) -> H (F,3,3) with uv ~ H [X, Y, 1]."""
    def norm(P):
        c = P.mean(-2, keepdims=True)
        s = np.sqrt(2.0) / np.sqrt(((P - c) ** 2).sum(-1).mean(-1))[..., na, na]
        T = np.zeros(P.shape[:-2] + (3, 3))
        T[..., 0, 0] = T[..., 1, 1] = s[..., 0, 0]
        T[..., 0, 2], T[..., 1, 2] = -s[..., 0, 0] * c[..., 0, 0], -s[..., 0, 0] * c[..., 0, 1]
        T[..., 2, 2] = 1
        return (P - c) * s, T

    uv = np.asarray(uv, dtype=np.float64)
    Xn, TX = norm(np.asarray(XY, dtype=np.float64))
    un, Tu = norm(uv)
    F, N = uv.shape[:2]
    X, Y = np.broadcast_to(Xn[:, 0], (F, N)), np.broadcast_to(Xn[:, 1], (F, N))
    u, v = un[..., 0], un[..., 1]
    z, o = np.zeros((F, N)), np.ones((F, N))
    A = np.concatenate([np.stack([X, Y, o, z, z, z, -u * X, -u * Y, -u], -1), np.stack([z, z, z, X, Y, o, -v * X, -v * Y, -v], -1)], axis=1)
    _, _, Vt = np.linalg.svd(A, full_matrices=False)   # (2N >= 9 rows: Vt is 9 x 9 either way; the 2N x 2N U is not needed)
    Hn = Vt[:, -1].reshape(F, 3, 3)
    H = np.linalg.inv(Tu) @ Hn @ TX
    return H / H[:, 2:3, 2:3]


def poses_from_homographies(H, K):
    """Board pose (F,6) from H = K [r1 r2 t] up to scale, rotation re-orthonormalised, board in front of the camera."""
    M = np.linalg.inv(K) @ H
    lam = 2.0 / (np.linalg.norm(M[:, :, 0], axis=1) + np.linalg.norm(M[:, :, 1], axis=1))
    lam = np.where(M[:, 2, 2] < 0, -lam, lam)  # t_z > 0
    M = M * lam[:, na, na]
    R = np.stack([M[:, :, 0], M[:, :, 1], np.cross(M[:, :, 0], M[:, :, 1])], axis=-1)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    neg = np.linalg.det(R) < 0
    if neg.any():
        U[neg, :, 2] *= -1
        R = U @ Vt
    return np.concatenate([rodrigues_inv(R), M[:, :, 2]], axis=-1)


def undistort_normalized(uv, K, dist, iterations=8):
    """Pixel -> undistorted normalised coordinates for OpenCV's (k1, k2, p1, p2, k3) model: the fixed-point iteration of its undistortPoints,
    x <- (x_d - tangential(x)) / radial(x)."""
    k1, k2, p1, p2, k3 = (list(np.ravel(dist)) + [0.0] * 5)[:5]
    xd = (uv[..., 0] - K[0, 2]) / K[0, 0]
    yd = (uv[..., 1] - K[1, 2]) / K[1, 1]
    x, y = xd.copy(), yd.copy()
    for _ in range(iterations):
        s = x * x + y * y
        d = 1 + s * (k1 + s * (k2 + s * k3))
        dx = 2 * p1 * x * y + p2 * (s + 2 * x * x)
        dy = p1 * (s + 2 * y * y) + 2 * p2 * x * y
        x, y = (xd - dx) / d, (yd - dy) / d
    return np.stack([x, y], -1)


def estimate_all_extrinsics(all_calib_poses, tree, root=0):
    """calibration.py:226-235 for a given spanning tree: chain the pairwise medians from the root (whose transform is the identity)."""
    ext = [None] * len(all_calib_poses)
    ext[root] = np.eye(4)
    for c1, c2 in tree:
        ext[c2] = get_transformation_matrix(estimate_pairwise_camera_transform(all_calib_poses[c1], all_calib_poses[c2])) @ ext[c1]
    return np.array([get_transformation_vector(T) for T in ext])


def solve_pnp(uv, obj, intr9, pose0):
    """The minimiser cv2.solvePnP (iterative) looks for: least squares of the pixel reprojection error of one view over its 6 pose coordinates,
    by scipy from the given start."""
    res = least_squares(lambda p: (np.asarray(uv, dtype=float) - project5(obj, p, intr9)).ravel(), np.asarray(pose0, dtype=float), xtol=1e-15, ftol=1e-15, gtol=1e-12)
    return res.x, res.cost


# ---- Zhang's closed form (the start cv2.calibrateCamera computes before it refines, calibration.py:68), numpy restatement: the checker of
# csrc/mcba_pnp_math.h: zhang_accumulate / zhang_solve (the product finds the null vector as an eigenvector of the 6 x 6 normal matrix on the
# GPU; here it is the last right singular vector of the stacked system).  Until round 6 this was the product's host code.
def _fallback_K(image_size):
    w, h = float(image_size[0]), float(image_size[1])
    return np.array([[max(w, h), 0, (w - 1) / 2], [0, max(w, h), (h - 1) / 2], [0, 0, 1.0]])


def intrinsics_from_homographies_batch(H, cams, image_sizes):
    """Zhang's closed form for every camera at once.  H (V,3,3): board-plane homographies of views; cams (V,) int: the camera of each view;
    image_sizes: one (width, height) per camera.  Returns K (C,3,3): the camera matrix from the image of the absolute conic (zero skew imposed),
    or the fallback f = max(w, h), c = image centre for a camera whose views do not constrain it (fewer than 2 usable views, or a non-positive-
    definite estimate).  One stacked SVD for all cameras (their rows padded with zeros to a common count: zero rows do not move a null vector)."""
    C = len(image_sizes)
    H = np.asarray(H, dtype=np.float64).reshape(-1, 3, 3)
    cams = np.asarray(cams, dtype=np.int64).reshape(-1)
    K = np.stack([_fallback_K(sz) for sz in image_sizes])
    ok = np.isfinite(H).all((1, 2))
    H, cams = H[ok], cams[ok]
    counts = np.bincount(cams, minlength=C)
    if not len(H) or counts.max() < 2:
        return K
    wh = np.asarray(image_sizes, dtype=np.float64).reshape(C, 2)
    s0 = wh.max(1)   # work in image coordinates of order 1 (pixel-scale entries would spread V over 12 decades): x' = (x - (w - 1) / 2) / s0
    ox, oy = (wh[:, 0] - 1) / 2, (wh[:, 1] - 1) / 2
    sv, oxv, oyv = (1 / s0)[cams, None], ox[cams, None], oy[cams, None]
    Hs = np.empty_like(H)
    Hs[:, 0] = (H[:, 0] - oxv * H[:, 2]) * sv
    Hs[:, 1] = (H[:, 1] - oyv * H[:, 2]) * sv
    Hs[:, 2] = H[:, 2]
    Hs /= np.sqrt(np.einsum("vij,vij->v", Hs[:, :, :2], Hs[:, :, :2]))[:, na, None]
    # Zhang's rows v_01 and v_00 - v_11 of every view, written straight into the cameras' stacked systems (views of a camera in list order; the
    # cameras' row counts padded with zeros to a common one: zero rows do not move a null vector)
    order = np.argsort(cams, kind="stable")
    cs = cams[order]
    slot = np.arange(len(cams)) - (np.cumsum(counts) - counts)[cs]   # position of each (sorted) view inside its camera
    V = np.zeros((C, max(2 * counts.max() + 1, 6), 6))   # (six rows at least: the reduced SVD of a 5 x 6 stack has no sixth right vector)
    a, b = Hs[order, :, 0], Hs[order, :, 1]
    r0, r1 = V[cs, 2 * slot], V[cs, 2 * slot + 1]   # (copies: fancy indexing), filled and written back
    r0[:, 0], r0[:, 1], r0[:, 2] = a[:, 0] * b[:, 0], a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0], a[:, 1] * b[:, 1]
    r0[:, 3], r0[:, 4], r0[:, 5] = a[:, 2] * b[:, 0] + a[:, 0] * b[:, 2], a[:, 2] * b[:, 1] + a[:, 1] * b[:, 2], a[:, 2] * b[:, 2]
    r1[:, 0], r1[:, 1], r1[:, 2] = a[:, 0] * a[:, 0] - b[:, 0] * b[:, 0], 2 * (a[:, 0] * a[:, 1] - b[:, 0] * b[:, 1]), a[:, 1] * a[:, 1] - b[:, 1] * b[:, 1]
    r1[:, 3], r1[:, 4], r1[:, 5] = 2 * (a[:, 2] * a[:, 0] - b[:, 2] * b[:, 0]), 2 * (a[:, 2] * a[:, 1] - b[:, 2] * b[:, 1]), a[:, 2] * a[:, 2] - b[:, 2] * b[:, 2]
    V[cs, 2 * slot], V[cs, 2 * slot + 1] = r0, r1
    V[:, -1, 1] = counts                                             # last row: skew = 0, weighted like the reference's single-camera form
    bvec = np.linalg.svd(V, full_matrices=False)[2][:, -1]          # (C, 6): b11 b12 b22 b13 b23 b33
    with np.errstate(all="ignore"):
        b11, b12, b22, b13, b23, b33 = bvec.T
        den = b11 * b22 - b12 * b12
        v0 = (b12 * b13 - b11 * b23) / den
        lam = b33 - (b13 * b13 + v0 * (b12 * b13 - b11 * b23)) / b11
        a2, b2 = lam / b11, lam * b11 / den
        good = (counts >= 2) & (den != 0) & (b11 != 0) & np.isfinite(a2) & np.isfinite(b2) & np.isfinite(v0) & (a2 > 0) & (b2 > 0)
        alpha, beta = np.sqrt(np.where(good, a2, 1.0)), np.sqrt(np.where(good, b2, 1.0))
        u0 = -b13 * alpha * alpha / lam
        # back to pixels: K = N^-1 K', N^-1 = [[s0, 0, ox], [0, s0, oy], [0, 0, 1]]
        Kg = np.zeros((C, 3, 3))
        Kg[:, 0, 0], Kg[:, 1, 1], Kg[:, 0, 2], Kg[:, 1, 2], Kg[:, 2, 2] = s0 * alpha, s0 * beta, s0 * u0 + ox, s0 * v0 + oy, 1.0
    K[good] = Kg[good]
    return K


def intrinsics_from_homographies(H, image_size):
    """K from the image of the absolute conic (zero skew imposed) for ONE camera's views H (V,3,3); falls back to f = max(w, h), c = image
    centre when the views do not constrain it (fewer than 2 usable views, or a non-positive-definite estimate)."""
    H = np.asarray(H, dtype=np.float64).reshape(-1, 3, 3)
    return intrinsics_from_homographies_batch(H, np.zeros(len(H), dtype=np.int64), [image_size])[0]


