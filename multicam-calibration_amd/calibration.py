"""`calibrate()` without OpenCV: the initialiser that produces `bundle_adjust`'s inputs.

Reference: multicam_calibration/calibration.py.  Same signatures, same return values, same use of the global numpy
RNG, same printed progress lines (the tqdm bars are not reproduced).  What differs is how the two OpenCV calls are
served -- this image has no cv2, and a GPU is there anyway:

  get_intrinsics (calibration.py:11-71, cv2.calibrateCamera with CALIB_FIX_K3 | CALIB_ZERO_TANGENT_DIST)
      closed-form start (Zhang 2000: one homography per view -> image of the absolute conic -> K; per-view pose from the
      homography) + joint refinement of (fx, fy, cx, cy, k1, k2) and all view poses by the library's own LM on the GPU:
      it is the bundle adjustment of ONE camera whose extrinsics are held at the identity, plain least squares.
  estimate_pose (calibration.py:74-113, cv2.solvePnP, iterative)
      homography start on undistorted normalised coordinates + the same GPU LM with ALL camera parameters held fixed
      (every frame is then an independent 6-parameter problem; the reduced camera system is the identity).
  the pose-graph part (calibration.py:116-277: pairwise medians, maximum spanning tree, chaining, consensus median)
      is numpy and is restated here; tests/golden/calibration_graph.npz pins it to the reference's own outputs
      (including networkx's tie-breaking in the spanning tree, reproduced without networkx).

  get_intrinsics(fix_k3=False | zero_tangent_dist=False), estimate_pose with tangential / k3 coefficients
      OpenCV's five-coefficient model (k1 k2 p1 p2 k3), which the bundle-adjustment kernels do not have: per-view normal equations
      from csrc/mcba_calib.hip (forward-mode automatic differentiation), the reduced 9 x 9 system and the damping here
      (_refine_five_coefficients).

Parity of the two OpenCV-backed pieces cannot be pinned to cv2 numbers in this container ("parity unpinned" for them);
they minimise the same reprojection error over the same parameters, and the tests check recovery of the synthetic truth
and that `bundle_adjust` started from `calibrate()` ends in the same optimum as from any other start.
"""
import numpy as np

na = np.newaxis


# ------------------------------------------------------------------ rigid transforms (geometry.py:8-56, 155-196)
def rodrigues(r):
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
    T[..., :3, :3] = rodrigues(t[..., :3])
    T[..., :3, 3] = t[..., 3:]
    T[..., 3, 3] = 1
    return T


def get_transformation_vector(T):
    return np.concatenate([rodrigues_inv(T[..., :3, :3]), T[..., :3, 3]], axis=-1)


# ------------------------------------------------------------------ pose graph (calibration.py:116-277)
def estimate_pairwise_camera_transform(camera1_poses, camera2_poses):
    """Median over the common frames of T2 T1^-1, component-wise on the 6-vectors (calibration.py:116-143)."""
    camera1_poses, camera2_poses = np.asarray(camera1_poses), np.asarray(camera2_poses)
    common = ~np.isnan([camera1_poses, camera2_poses]).any((0, 2))
    T1 = get_transformation_matrix(camera1_poses[common])
    T2 = get_transformation_matrix(camera2_poses[common])
    return np.median(get_transformation_vector(T2 @ np.linalg.inv(T1)), axis=0)


def get_camera_spanning_tree(all_calib_poses, root=0):
    """Maximum spanning tree of the co-detection graph, edges ordered by distance from `root` (calibration.py:146-197).

    The reference delegates to networkx (Kruskal on a stably sorted edge list, then `Graph.edges` iteration order); both
    are reproduced here so that ties between equally populated camera pairs resolve identically."""
    all_calib_poses = np.asarray(all_calib_poses)
    C = len(all_calib_poses)
    detected = ~np.isnan(all_calib_poses).any(2)
    edges = [(i, j, int((detected[i] & detected[j]).sum())) for i in range(C) for j in range(i + 1, C)]
    parent = list(range(C))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    adj = [[] for _ in range(C)]  # neighbours in the order the tree edges were accepted (= networkx adjacency order)
    for i, j, _ in sorted(edges, key=lambda e: e[2], reverse=True):  # stable: equal weights keep (i, j) order
        a, b = find(i), find(j)
        if a != b:
            parent[a] = b
            adj[i].append(j)
            adj[j].append(i)
    seen, tree = set(), []  # networkx Graph.edges: nodes in insertion order, each edge reported from its first endpoint
    for u in range(C):
        for v in adj[u]:
            if v not in seen:
                tree.append((u, v))
        seen.add(u)
    dist = {root: 0}
    frontier = [root]
    while frontier:
        nxt = []
        for u in frontier:
            for v in adj[u]:
                if v not in dist:
                    dist[v] = dist[u] + 1
                    nxt.append(v)
        frontier = nxt
    tree = [tuple(sorted(e, key=lambda n: dist[n])) for e in tree]  # KeyError if the graph is disconnected, as the reference
    return sorted(tree, key=lambda e: dist[e[0]])


def estimate_all_extrinsics(all_calib_poses, root=0):
    """Chain the pairwise transforms down the spanning tree (calibration.py:200-236)."""
    all_calib_poses = np.asarray(all_calib_poses)
    ext = [None] * len(all_calib_poses)
    ext[root] = np.eye(4)
    tree = get_camera_spanning_tree(all_calib_poses, root=root)
    for c1, c2 in tree:
        ext[c2] = get_transformation_matrix(estimate_pairwise_camera_transform(all_calib_poses[c1], all_calib_poses[c2])) @ ext[c1]
    return np.array([get_transformation_vector(T) for T in ext]), tree


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


# ------------------------------------------------------------------ closed-form starts (Zhang 2000, sections 3.1, 3.2 and appendix A)
def _require_planar(obj):
    obj = np.asarray(obj, dtype=np.float64)
    if obj.ndim != 2 or obj.shape[1] != 3 or np.abs(obj[:, 2]).max() > 1e-9 * max(1.0, np.abs(obj[:, :2]).max()):
        raise NotImplementedError("the closed-form initialisation needs a planar calibration board (z = 0), as the reference's boards are")
    return obj


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
    _, _, Vt = np.linalg.svd(A)
    Hn = Vt[:, -1].reshape(F, 3, 3)
    H = np.linalg.inv(Tu) @ Hn @ TX
    return H / H[:, 2:3, 2:3]


def intrinsics_from_homographies(H, image_size):
    """K from the image of the absolute conic (zero skew imposed); falls back to f = max(w, h), c = image centre when
    the views do not constrain it (fewer than 2 usable views, or a non-positive-definite estimate)."""
    w, h = float(image_size[0]), float(image_size[1])
    fallback = np.array([[max(w, h), 0, (w - 1) / 2], [0, max(w, h), (h - 1) / 2], [0, 0, 1.0]])
    if len(H) < 2:
        return fallback

    def vij(H, i, j):
        a, b = H[:, :, i], H[:, :, j]
        return np.stack([a[:, 0] * b[:, 0], a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0], a[:, 1] * b[:, 1],
                         a[:, 2] * b[:, 0] + a[:, 0] * b[:, 2], a[:, 2] * b[:, 1] + a[:, 1] * b[:, 2], a[:, 2] * b[:, 2]], -1)

    s0 = max(w, h)  # work in image coordinates of order 1 (pixel-scale entries would spread V over 12 decades)
    Nrm = np.array([[1 / s0, 0, -(w - 1) / (2 * s0)], [0, 1 / s0, -(h - 1) / (2 * s0)], [0, 0, 1.0]])
    Hs = Nrm @ H
    Hs = Hs / np.linalg.norm(Hs[:, :, :2], axis=(1, 2), keepdims=True)
    V = np.concatenate([vij(Hs, 0, 1), vij(Hs, 0, 0) - vij(Hs, 1, 1), np.array([[0, 1.0, 0, 0, 0, 0]]) * len(H)])  # last row: skew = 0
    _, _, Vt = np.linalg.svd(V)
    b11, b12, b22, b13, b23, b33 = Vt[-1]
    den = b11 * b22 - b12 * b12
    if den == 0 or b11 == 0:
        return fallback
    v0 = (b12 * b13 - b11 * b23) / den
    lam = b33 - (b13 * b13 + v0 * (b12 * b13 - b11 * b23)) / b11
    a2, b2 = lam / b11, lam * b11 / den
    if not (np.isfinite([a2, b2, v0]).all() and a2 > 0 and b2 > 0):
        return fallback
    alpha, beta = np.sqrt(a2), np.sqrt(b2)
    u0 = -b13 * alpha * alpha / lam
    K = np.linalg.inv(Nrm) @ np.array([[alpha, 0, u0], [0, beta, v0], [0, 0, 1.0]])
    return K / K[2, 2]


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


# ------------------------------------------------------------------ GPU refinement: the library's LM with a parameter mask
def _refine_single_camera(uvs, obj, cam12, poses, free_cam, device, max_nfev=200):
    from . import ops, solver

    prob = ops.Problem(np.ascontiguousarray(uvs[na]), obj, device=device, loss="linear")
    try:
        x0 = np.concatenate([cam12, np.asarray(poses, dtype=np.float64).ravel()])
        res = solver.lm_solve(prob, x0, ftol=1e-12, xtol=1e-12, gtol=1e-10, max_nfev=max_nfev, verbose=0, free_cam_mask=free_cam)
    finally:
        prob.close()
    return res.x[:12], res.x[12:].reshape(-1, 6), res


# ------------------------------------------------------------------ the five-coefficient model (k1 k2 p1 p2 k3): get_intrinsics(fix_k3=False | zero_tangent_dist=False)
def _refine_five_coefficients(uvs, obj, intr9, poses, free9, device, max_evaluations=300):
    """Levenberg-Marquardt on calibrateCamera's objective (sum of squared reprojection errors over the views) with OpenCV's five-coefficient
    model: the per-view Gauss-Newton blocks, gradients and costs come from the GPU (ops.calib_normal_equations: forward-mode automatic
    differentiation, one wavefront per view); the Schur complement over the views' 6 x 6 pose blocks and the damping live here (a 9 x 9 system).
    free9: which of fx fy cx cy k1 k2 p1 p2 k3 are variables; none = every view is an independent 6-parameter problem (estimate_pose) with
    its own damping.  Returns (intr9, poses, cost)."""
    from . import ops

    k, ps = np.array(intr9, dtype=np.float64), np.array(poses, dtype=np.float64)
    free9 = np.asarray(free9, dtype=bool)
    nf, V = int(free9.sum()), len(ps)
    H, g, c = ops.calib_normal_equations(uvs, obj, k, ps, device)
    if not np.isfinite(c).all():
        raise ValueError("Residuals are not finite in the initial point.")
    per_view = nf == 0
    lam, nu = (np.full(V, 1e-3), np.full(V, 2.0)) if per_view else (1e-3, 2.0)
    done = np.zeros(V, dtype=bool)
    eye6 = np.eye(6)
    for _ in range(max_evaluations):
        Vv, gv = H[:, 9:, 9:], g[:, 9:]
        Dv = np.maximum(np.einsum("vii->vi", Vv), 1e-300)
        lam_v = lam if per_view else np.full(V, lam)
        Vinv = np.linalg.inv(Vv + (lam_v[:, None] * Dv)[:, :, None] * eye6)
        dc = np.zeros(9)
        if nf:
            U, gc = H[:, :9, :9].sum(0)[np.ix_(free9, free9)], g[:, :9].sum(0)[free9]
            W = H[:, :9, 9:][:, free9, :]                                   # (V, nf, 6)
            WVi = W @ Vinv
            S = U + lam * np.diag(np.maximum(np.diag(U), 1e-300)) - (WVi @ W.transpose(0, 2, 1)).sum(0)
            try:
                dc[free9] = np.linalg.solve(S, -gc + np.einsum("vij,vj->i", WVi, gv))
            except np.linalg.LinAlgError:   # (views that do not constrain the intrinsics: more damping)
                lam, nu = lam * nu, nu * 2.0
                if lam > 1e12:
                    break
                continue
            dv = -np.einsum("vij,vj->vi", Vinv, gv + np.einsum("vij,i->vj", W, dc[free9]))
        else:
            dv = -np.einsum("vij,vj->vi", Vinv, gv)
        dv[done] = 0.0
        d15 = np.concatenate([np.broadcast_to(dc, (V, 9)), dv], axis=1)
        pred_v = -np.einsum("vi,vi->v", g, d15) - 0.5 * np.einsum("vi,vij,vj->v", d15, H, d15)   # per view; their sum is the model's reduction
        H2, g2, c2 = ops.calib_normal_equations(uvs, obj, k + dc, ps + dv, device)
        if per_view:
            gain = c - c2
            ok = np.isfinite(c2) & (gain >= 0) & ~done
            ratio = np.where(pred_v > 0, gain / np.where(pred_v > 0, pred_v, 1.0), -1.0)
            conv = ok & (gain <= 1e-14 * np.maximum(c, 1e-300))
            ps[ok], H[ok], g[ok], c[ok] = (ps + dv)[ok], H2[ok], g2[ok], c2[ok]
            lam = np.where(ok, np.maximum(lam * np.maximum(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3), 1e-12), lam * nu)
            nu = np.where(ok, 2.0, nu * 2.0)
            done |= conv | (lam > 1e12)
            if done.all():
                break
        else:
            gain, pred = c.sum() - c2.sum(), pred_v.sum()
            if np.isfinite(c2).all() and gain >= 0:
                ratio = gain / pred if pred > 0 else 1.0
                small = gain <= 1e-14 * max(c.sum(), 1e-300)
                k, ps, H, g, c = k + dc, ps + dv, H2, g2, c2
                lam, nu = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3), 1e-12), 2.0
                if small:
                    break
            else:
                lam, nu = lam * nu, nu * 2.0
                if lam > 1e12:
                    break
    return k, ps, float(c.sum())


def _zhang_start(uvs, obj, image_size):
    H = homographies(obj[:, :2], uvs)
    K0 = intrinsics_from_homographies(H, image_size)
    return K0, poses_from_homographies(H, K0)


def get_intrinsics(calib_uvs, calib_objpoints, image_size, n_samples=100, fix_k3=True, zero_tangent_dist=True, device=0):
    """Camera matrix (3,3) and distortion (k1, k2, p1, p2, k3) from complete detections of a planar board (calibration.py:11-71: the model
    of cv2.calibrateCamera under flags = CALIB_FIX_K3 * fix_k3 + CALIB_ZERO_TANGENT_DIST * zero_tangent_dist).  The reference's default --
    both flags: the (k1, k2) radial model, which is also all bundle_adjust optimises -- is refined by the library's own bundle-adjustment
    loop; with a flag off, the five-coefficient model by _refine_five_coefficients (GPU normal equations, 9 x 9 reduced system here)."""
    obj = _require_planar(calib_objpoints)
    if not (fix_k3 and zero_tangent_dist):
        calib_uvs = np.asarray(calib_uvs, dtype=np.float64)
        calib_uvs = calib_uvs[~np.isnan(calib_uvs).any((1, 2))]
        n_samples = min(n_samples, len(calib_uvs))
        if n_samples < 1:
            raise ValueError("no complete detection of the calibration board for this camera")
        uvs = calib_uvs[np.random.choice(len(calib_uvs), n_samples, replace=False)]  # same draw from the global RNG as the reference
        K0, poses0 = _zhang_start(uvs, obj, image_size)
        free9 = np.array([1, 1, 1, 1, 1, 1, not zero_tangent_dist, not zero_tangent_dist, not fix_k3], dtype=bool)
        k9, _, _ = _refine_five_coefficients(uvs, obj, [K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], poses0, free9, device)
        K = np.array([[k9[0], 0, k9[2]], [0, k9[1], k9[3]], [0, 0, 1.0]])
        return K, k9[4:].copy()
    calib_uvs = np.asarray(calib_uvs, dtype=np.float64)
    calib_uvs = calib_uvs[~np.isnan(calib_uvs).any((1, 2))]
    n_samples = min(n_samples, len(calib_uvs))
    if n_samples < 1:
        raise ValueError("no complete detection of the calibration board for this camera")
    uvs = calib_uvs[np.random.choice(len(calib_uvs), n_samples, replace=False)]  # same draw from the global RNG as the reference
    H = homographies(obj[:, :2], uvs)
    K0 = intrinsics_from_homographies(H, image_size)
    poses0 = poses_from_homographies(H, K0)
    cam0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0.0, 0.0, 0, 0, 0, 0, 0, 0])
    free = np.r_[np.ones(6, bool), np.zeros(6, bool)]  # the camera IS the coordinate frame
    cam, _, _ = _refine_single_camera(uvs, obj, cam0, poses0, free, device)
    K = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1.0]])
    return K, np.array([cam[4], cam[5], 0.0, 0.0, 0.0])


def estimate_pose(calib_uvs, calib_objpoints, camera_matrix, dist_coeffs, device=0):
    """Board pose (board -> camera) per frame, NaN rows where the detection is incomplete (calibration.py:74-113)."""
    obj = _require_planar(calib_objpoints)
    calib_uvs = np.asarray(calib_uvs, dtype=np.float64)
    K = np.asarray(camera_matrix, dtype=np.float64)
    dist = np.zeros(5)
    dist[: np.size(dist_coeffs)] = np.ravel(dist_coeffs)
    poses = np.full((len(calib_uvs), 6), np.nan)
    ok = ~np.isnan(calib_uvs).any((1, 2))
    if not ok.any():
        return poses
    uvs = calib_uvs[ok]
    H = homographies(obj[:, :2], undistort_normalized(uvs, K, dist))
    poses0 = poses_from_homographies(H, np.eye(3))
    if dist[2] != 0.0 or dist[3] != 0.0 or dist[4] != 0.0:   # tangential / k3 coefficients (get_intrinsics with a flag off): the five-coefficient model
        _, refined, _ = _refine_five_coefficients(uvs, obj, [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *dist], poses0, np.zeros(9, bool), device)
    else:
        cam = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], dist[0], dist[1], 0, 0, 0, 0, 0, 0])
        _, refined, _ = _refine_single_camera(uvs, obj, cam, poses0, np.zeros(12, bool), device)
    poses[ok] = refined
    return poses


def calibrate(all_calib_uvs, img_sizes, calib_objpoints, root=0, verbose=True, n_samples_for_intrinsics=100, device=0):
    """Reference signature and return tuple (calibration.py:280-373):
    (all_extrinsics (C,6), all_intrinsics [(K, dist5)] * C, consensus board poses (F,6), spanning_tree)."""
    all_intrinsics = []
    if verbose:
        print("Estimating camera intrinsics")
    for uvs, img_size in zip(all_calib_uvs, img_sizes):
        all_intrinsics.append(get_intrinsics(uvs, calib_objpoints, img_size, n_samples=n_samples_for_intrinsics, device=device))
    if verbose:
        print("Initializing calibration object poses")
    all_calib_poses = np.array([estimate_pose(uvs, calib_objpoints, *intr, device=device) for uvs, intr in zip(all_calib_uvs, all_intrinsics)])
    if verbose:
        print("Estimating camera extrinsics")
    all_extrinsics, spanning_tree = estimate_all_extrinsics(all_calib_poses, root=root)
    if verbose:
        print("Merging calibration object poses")
    calib_poses = consensus_calib_poses(all_calib_poses, all_extrinsics)
    return all_extrinsics, all_intrinsics, calib_poses, spanning_tree
