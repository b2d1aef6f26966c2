"""`calibrate()` without OpenCV: the initialiser that produces `bundle_adjust`'s inputs.

Reference: multicam_calibration/calibration.py.  Same signatures, same return values, same use of the global numpy
RNG, same printed progress lines (the tqdm bars are not reproduced).  This image has no cv2, and a GPU is there anyway: everything
per view and per frame runs in libmcba.so (include/mcba.h: the mcba_calib_* block, csrc/mcba_pnp.hip); the host keeps what is a
handful of numbers per CAMERA (the reference's RNG draw, the spanning tree, the chaining of C - 1 transforms).

  get_intrinsics (calibration.py:11-71, cv2.calibrateCamera with CALIB_FIX_K3 | CALIB_ZERO_TANGENT_DIST)
      per sampled view the board-plane homography (GPU: normalised DLT, lane = view) -> Zhang's closed form for K (GPU: the null vector
      of a 6-column system per camera, by cyclic Jacobi on its normal matrix) -> per-view pose (GPU: cv2.solvePnP's job with K0) -- these
      three in one crossing, mcba_calib_start -> joint refinement of (fx, fy, cx, cy, k1, k2)
      and the views' poses by the library's own device-resident LM loop: the bundle adjustment of cameras whose extrinsics are held at
      the identity, plain least squares.  calibrate() refines EVERY camera in one such run (the sampled views of all cameras side by
      side: the normal equations are block-diagonal over the cameras).
  estimate_pose (calibration.py:74-113, cv2.solvePnP, iterative)
      ONE kernel launch for every (camera, frame): undistort, homography on normalised coordinates, pose from it, Levenberg-Marquardt
      on the pixel reprojection error with the five-coefficient model, every view its own damping and stopping test.
  the pose-graph part (calibration.py:116-277: pairwise medians, maximum spanning tree, chaining, consensus median)
      pairwise transforms and their exact medians (radix select) and the consensus median over cameras on the GPU; the spanning tree
      (C x C co-detection counts; networkx's tie-breaking reproduced without networkx) and the chaining of C - 1 transforms on the host.
      tests/golden/calibration_graph.npz pins all of it to the reference's own outputs.

  get_intrinsics(fix_k3=False | zero_tangent_dist=False)
      OpenCV's five-coefficient model (k1 k2 p1 p2 k3), which the bundle-adjustment kernels do not have: per-view normal equations
      from csrc/mcba_calib.hip (forward-mode automatic differentiation), the reduced 9 x 9 system and the damping here
      (_refine_five_coefficients).

Parity of the two OpenCV-backed pieces cannot be pinned to cv2 numbers in this container ("parity unpinned" for them);
they minimise the same reprojection error over the same parameters, and the tests check the kernels against numpy restatements
(oracle/calibration_oracle.py), recovery of the synthetic truth and that `bundle_adjust` started from `calibrate()` ends in the same
optimum as from any other start.
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
def estimate_pairwise_camera_transform(camera1_poses, camera2_poses, device=0):
    """Median over the common frames of T2 T1^-1, component-wise on the 6-vectors (calibration.py:116-143).  On the GPU: one lane per
    frame, exact medians by radix select (include/mcba.h: mcba_pose_pairwise)."""
    from . import ops

    poses = np.stack([np.asarray(camera1_poses, dtype=np.float64), np.asarray(camera2_poses, dtype=np.float64)])
    return ops.pose_pairwise(poses, [(0, 1)], device)[0][0]


def _co_detection_counts(detected):
    """(C,C) int64: frames in which both cameras have a pose = detected @ detected.T, by population counts of the AND of the bit-packed rows
    (an int64 matrix product of (C,F) flags does not go through BLAS: 0.25 ms at 6 x 10 000 against 0.02)."""
    bits = np.packbits(detected, axis=1)
    buf = np.zeros((bits.shape[0], (bits.shape[1] + 7) // 8 * 8), dtype=np.uint8)
    buf[:, : bits.shape[1]] = bits
    words = buf.view(np.uint64)
    return np.bitwise_count(words[:, None, :] & words[None, :, :]).sum(2, dtype=np.int64)


def _spanning_tree(detected, root=0):
    """Maximum spanning tree of the co-detection graph from the (C,F) detection flags, edges ordered by distance from `root`.

    The reference delegates to networkx (Kruskal on a stably sorted edge list, then `Graph.edges` iteration order); both
    are reproduced here so that ties between equally populated camera pairs resolve identically."""
    detected = np.asarray(detected, dtype=bool)
    C = len(detected)
    counts = _co_detection_counts(detected)
    edges = [(i, j, int(counts[i, j])) for i in range(C) for j in range(i + 1, C)]
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


def get_camera_spanning_tree(all_calib_poses, root=0):
    """Maximum spanning tree of the co-detection graph, edges ordered by distance from `root` (calibration.py:146-197)."""
    return _spanning_tree(~np.isnan(np.asarray(all_calib_poses)).any(2), root=root)


def _chain_extrinsics(n_cameras, tree, transforms, root):
    """World -> camera 6-vectors from the tree's pairwise transforms (calibration.py:226-235); the root's is the exact zero vector."""
    ext = np.zeros((n_cameras, 4, 4))
    ext[root] = np.eye(4)
    if len(tree):
        steps = get_transformation_matrix(np.asarray(transforms, dtype=np.float64).reshape(len(tree), 6))   # one call for every edge
        for (c1, c2), T in zip(tree, steps):
            ext[c2] = T @ ext[c1]
    with np.errstate(invalid="ignore"):
        return get_transformation_vector(ext)


def estimate_all_extrinsics(all_calib_poses, root=0, device=0):
    """Chain the pairwise transforms down the spanning tree (calibration.py:200-236)."""
    from . import ops

    all_calib_poses = np.asarray(all_calib_poses, dtype=np.float64)
    tree = get_camera_spanning_tree(all_calib_poses, root=root)
    transforms = ops.pose_pairwise(all_calib_poses, tree, device)[0] if tree else []
    return _chain_extrinsics(len(all_calib_poses), tree, transforms, root), tree


def consensus_calib_poses(all_calib_poses, all_extrinsics, device=0):
    """Per-camera board poses mapped to world coordinates, nan-median over cameras (calibration.py:239-277): lane = frame on the GPU."""
    from . import ops

    return ops.pose_consensus(np.asarray(all_calib_poses, dtype=np.float64), all_extrinsics, device)


# ------------------------------------------------------------------ closed-form start of the intrinsics (Zhang 2000, section 3.1 and appendix B)
def _require_planar(obj):
    obj = np.asarray(obj, dtype=np.float64)
    if obj.ndim != 2 or obj.shape[1] != 3 or np.abs(obj[:, 2]).max() > 1e-9 * max(1.0, np.abs(obj[:, :2]).max()):
        raise NotImplementedError("the closed-form initialisation needs a planar calibration board (z = 0), as the reference's boards are")
    obj = obj.copy()
    obj[:, 2] = 0.0
    return obj


def _intr9(K, dist=None):
    d = np.zeros(5)
    if dist is not None:
        d[: np.size(dist)] = np.ravel(dist)
    return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], *d])


def _draw_samples(complete, n_samples):
    """The reference's draw (calibration.py:55-60): one np.random.choice without replacement over the camera's complete detections, from
    the GLOBAL numpy RNG.  complete: (F,) bool.  Returns the sampled frame indices in draw order."""
    frames = np.flatnonzero(complete)
    n = min(n_samples, len(frames))
    if n < 1:
        raise ValueError("no complete detection of the calibration board for this camera")
    return frames[np.random.choice(len(frames), n, replace=False)]


def _start_on_device(prob, views, image_sizes):
    """Closed-form start for the cameras of `prob` from their sampled views ((V,2) int: camera, frame), ONE crossing (include/mcba.h:
    mcba_calib_start): per-view homographies, Zhang's K per camera from them (image of the absolute conic, zero skew; the fallback f = max(w, h),
    c = the image centre for a camera its views do not constrain), per-view poses with that K.  Returns (K0 list, poses0 (V,6))."""
    # (four linearisations per view: K0 is a closed-form estimate without distortion -- poses converged against it to 1e-13 are not better starts;
    #  the joint refinement takes the same number of evaluations from poses after 1, 4 or 60, scripts/start_evals_probe.py)
    k4, poses0 = prob.calib_start(views, np.asarray(image_sizes, dtype=np.float64).reshape(prob.C, 2), max_evaluations=4)
    return [np.array([[q[0], 0, q[2]], [0, q[1], q[3]], [0, 0, 1.0]]) for q in k4], poses0


def _refine_intrinsics_on_device(prob, views, K0, poses0):
    """get_intrinsics' refinement for every camera of `prob` at once: the library's device-resident LM loop on the sampled views side by
    side (ops.Problem.view_subset), camera extrinsics held at the identity, plain least squares.  Returns cam (C,12)."""
    from . import solver

    ok = np.isfinite(poses0).all(1)
    views, poses0 = views[ok], poses0[ok]
    cam0 = np.zeros((prob.C, 12))
    for c, K in enumerate(K0):
        cam0[c, :4] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    if not len(views):
        return cam0
    free = np.tile(np.r_[np.ones(6, bool), np.zeros(6, bool)], prob.C)  # each camera IS the coordinate frame of its own views
    free.reshape(prob.C, 12)[~np.isin(np.arange(prob.C), views[:, 0])] = False   # (a camera without a usable view keeps its start)
    sub = prob.view_subset(views, loss="linear")
    try:
        # (1e-9: the cost is then stationary to 2e-10 of itself and the intrinsics to 1e-7 px -- next to the 4-9 px a hundred views leave them
        #  uncertain by; 1e-12 costs five more evaluations of 67 us for the same numbers: scripts/joint_lm_probe.py)
        res = solver.lm_solve(sub, np.concatenate([cam0.ravel(), poses0.ravel()]), ftol=1e-9, xtol=1e-9, gtol=1e-10, max_nfev=200, verbose=0, free_cam_mask=free)
    finally:
        sub.close()
    return res.x[: 12 * prob.C].reshape(prob.C, 12)


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
    flat = 0
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
            if np.isfinite(c2).all() and gain >= -1e-13 * c.sum():   # (a loss inside the cost's own rounding level is not an uphill step)
                ratio = gain / pred if pred > 0 else 1.0
                # a gain at the rounding level of the cost does not mean the weakly determined coefficients (k2 against k3) have settled: the
                # step is computed from the gradient, which is far more exact than a difference of costs -- stop on the step, or after a
                # few such steps in a row
                flat = flat + 1 if gain <= 1e-14 * max(c.sum(), 1e-300) else 0
                settled = np.max(np.abs(dc[free9]) / (np.abs(k[free9]) + 1e-4)) <= 1e-9 if nf else True
                k, ps, H, g, c = k + dc, ps + dv, H2, g2, c2
                lam, nu = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3), 1e-12), 2.0
                if flat and (settled or flat >= 6):
                    break
            else:
                lam, nu = lam * nu, nu * 2.0
                if lam > 1e12:
                    break
    return k, ps, float(c.sum())


def _zhang_start(uvs, obj, image_size, device=0):
    """Closed-form start (K0, per-view poses) from complete views uvs (V,N,2) of one camera."""
    from . import ops

    prob = ops.Problem(np.ascontiguousarray(uvs[na]), obj, device=device, loss="linear")
    try:
        views = np.stack([np.zeros(len(uvs), np.int32), np.arange(len(uvs), dtype=np.int32)], 1)
        K0, poses0 = _start_on_device(prob, views, [image_size])
    finally:
        prob.close()
    return K0[0], poses0


def get_intrinsics(calib_uvs, calib_objpoints, image_size, n_samples=100, fix_k3=True, zero_tangent_dist=True, device=0):
    """Camera matrix (3,3) and distortion (k1, k2, p1, p2, k3) from complete detections of a planar board (calibration.py:11-71: the model
    of cv2.calibrateCamera under flags = CALIB_FIX_K3 * fix_k3 + CALIB_ZERO_TANGENT_DIST * zero_tangent_dist).  The reference's default --
    both flags: the (k1, k2) radial model, which is also all bundle_adjust optimises -- is refined by the library's own bundle-adjustment
    loop; with a flag off, the five-coefficient model by _refine_five_coefficients (GPU normal equations, 9 x 9 reduced system here)."""
    from . import ops

    obj = _require_planar(calib_objpoints)
    calib_uvs = np.ascontiguousarray(calib_uvs, dtype=np.float64)
    prob = ops.Problem(calib_uvs[na], obj, device=device, loss="linear")
    try:
        frames = _draw_samples(prob.calib_complete()[0], n_samples)  # same draw from the global RNG as the reference
        views = np.stack([np.zeros(len(frames), np.int32), frames.astype(np.int32)], 1)
        K0, poses0 = _start_on_device(prob, views, [image_size])
        if fix_k3 and zero_tangent_dist:
            cam = _refine_intrinsics_on_device(prob, views, K0, poses0)[0]
            return np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1.0]]), np.array([cam[4], cam[5], 0.0, 0.0, 0.0])
    finally:
        prob.close()
    ok = np.isfinite(poses0).all(1)
    free9 = np.array([1, 1, 1, 1, 1, 1, not zero_tangent_dist, not zero_tangent_dist, not fix_k3], dtype=bool)
    k9, _, _ = _refine_five_coefficients(calib_uvs[frames[ok]], obj, _intr9(K0[0]), poses0[ok], free9, device)
    K = np.array([[k9[0], 0, k9[2]], [0, k9[1], k9[3]], [0, 0, 1.0]])
    return K, k9[4:].copy()


def estimate_pose(calib_uvs, calib_objpoints, camera_matrix, dist_coeffs, device=0):
    """Board pose (board -> camera) per frame, NaN rows where the detection is incomplete (calibration.py:74-113).  One kernel launch
    (include/mcba.h: mcba_calib_poses); tangential / k3 coefficients (get_intrinsics with a flag off) are part of its model."""
    from . import ops

    obj = _require_planar(calib_objpoints)
    calib_uvs = np.ascontiguousarray(calib_uvs, dtype=np.float64)
    prob = ops.Problem(calib_uvs[na], obj, device=device, loss="linear")
    try:
        _, poses, _ = prob.calib_poses(_intr9(np.asarray(camera_matrix, dtype=np.float64), dist_coeffs)[na], want_poses=True)
    finally:
        prob.close()
    return poses[0]


def _sample_all_cameras(complete, n_samples):
    """(V,2) int32 (camera, frame): every camera's sampled views, drawn camera by camera as the reference's loop does (calibration.py:341-350;
    nothing else consumes the global RNG in between)."""
    return np.concatenate([np.stack([np.full(len(fr), c, np.int32), fr.astype(np.int32)], 1) for c, fr in ((c, _draw_samples(row, n_samples)) for c, row in enumerate(complete))])


def _pose_graph_on_device(prob, detected, root):
    """calibration.py:200-277 on the poses prob.calib_poses left on the device: the spanning tree here (C x C counts), the tree's pairwise medians,
    their chain from the root and the consensus in one crossing.  Returns (all_extrinsics, consensus poses, spanning tree)."""
    tree = _spanning_tree(detected, root=root)
    ext, poses = prob.calib_graph(tree, root)
    return ext, poses, tree


def calibrate(all_calib_uvs, img_sizes, calib_objpoints, root=0, verbose=True, n_samples_for_intrinsics=100, device=0):
    """Reference signature and return tuple (calibration.py:280-373):
    (all_extrinsics (C,6), all_intrinsics [(K, dist5)] * C, consensus board poses (F,6), spanning_tree).
    The detections go to the GPU once; every stage works on that copy (module docstring)."""
    from . import ops

    obj = _require_planar(calib_objpoints)
    prob = ops.Problem(np.ascontiguousarray(all_calib_uvs, dtype=np.float64), obj, device=device, loss="linear")
    try:
        if verbose:
            print("Estimating camera intrinsics")
        views = _sample_all_cameras(prob.calib_complete(), n_samples_for_intrinsics)
        K0, poses0 = _start_on_device(prob, views, img_sizes)
        cam = _refine_intrinsics_on_device(prob, views, K0, poses0)
        all_intrinsics = [(np.array([[q[0], 0, q[2]], [0, q[1], q[3]], [0, 0, 1.0]]), np.array([q[4], q[5], 0.0, 0.0, 0.0])) for q in cam]
        if verbose:
            print("Initializing calibration object poses")
        detected, _, _ = prob.calib_poses(np.array([_intr9(K, d) for K, d in all_intrinsics]))
        if verbose:
            print("Estimating camera extrinsics")
        all_extrinsics, calib_poses, spanning_tree = _pose_graph_on_device(prob, detected, root)
        if verbose:
            print("Merging calibration object poses")
    finally:
        prob.close()
    return all_extrinsics, all_intrinsics, calib_poses, spanning_tree
