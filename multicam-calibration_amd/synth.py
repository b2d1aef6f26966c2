"""Deterministic synthetic calibration-board detections.

Data generation only -- the few lines of numpy projection below exist to *make*
inputs (exact detections + noise) for tests and `bench.py`; they are not a compute
path of the solver and nothing in `solver.py`/`ops.py` calls them.

Board: rows x cols grid, `pitch` mm, z = 0 (same construction as the reference's
`generate_chessboard_objpoints`, detection.py:492-518, cast to float64).
Cameras: ring of radius 600 mm, 400 mm above the board origin, looking at it.
World frame = camera 0's frame, so camera 0's extrinsic is the exact zero 6-vector,
as `calibrate()` returns it (calibration.py:200-236) -- the theta = 0 case of the
rotation-vector Jacobian is therefore exercised by every synthetic problem.
"""
import numpy as np


def board_points(rows=6, cols=9, pitch=12.5):
    g = np.mgrid[0:rows, 0:cols].T.reshape(-1, 2).astype(np.float64) * pitch
    return np.concatenate([g, np.zeros((g.shape[0], 1))], axis=1)


def _rot(r):
    r = np.asarray(r, dtype=np.float64)
    th = np.linalg.norm(r, axis=-1)[..., None, None]
    K = np.zeros(r.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -r[..., 2], r[..., 1]
    K[..., 1, 0], K[..., 1, 2] = r[..., 2], -r[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -r[..., 1], r[..., 0]
    th_safe = np.where(th == 0, 1.0, th)
    a = np.where(th == 0, 1.0, np.sin(th) / th_safe)
    b = np.where(th == 0, 0.5, (1 - np.cos(th)) / th_safe**2)
    return np.eye(3) + a * K + b * (K @ K)


def _rotvec(R):
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], axis=-1)
    th = np.arccos(np.clip((np.trace(R, axis1=-2, axis2=-1) - 1) / 2, -1, 1))[..., None]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v * th / (n + (n == 0))


def _T(t6):
    T = np.zeros(t6.shape[:-1] + (4, 4))
    T[..., :3, :3] = _rot(t6[..., :3])
    T[..., :3, 3] = t6[..., 3:]
    T[..., 3, 3] = 1
    return T


def _t6(T):
    return np.concatenate([_rotvec(T[..., :3, :3]), T[..., :3, 3]], axis=-1)


def project(cam12, poses, obj):
    """(C,F,N,2) exact detections: pinhole + (k1,k2) radial through the two chained rigid transforms."""
    Xw = np.einsum("fij,nj->fni", _rot(poses[:, :3]), obj) + poses[:, None, 3:]
    out = []
    for fx, fy, cx, cy, k1, k2, *ext in cam12:
        ext = np.asarray(ext)
        Xc = np.einsum("ij,fnj->fni", _rot(ext[:3]), Xw) + ext[3:]
        a, b = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
        s = a * a + b * b
        d = 1 + k1 * s + k2 * s * s
        out.append(np.stack([fx * a * d + cx, fy * b * d + cy], -1))
    return np.stack(out)


def make_problem(n_cameras, n_frames, rows=6, cols=9, pitch=12.5, seed=0, perturb_seed=1, noise=0.2, missing=0.0, outlier_frames=0, scalar_nans=0, frame_seed=None):
    """Returns a dict:
      uvs (C,F,N,2) f64 with NaN = missing, obj (N,3),
      true_cam (C,12), true_poses (F,6),
      extrinsics (C,6), intrinsics [(K 3x3, dist 5)]*C, poses (F,6)  -- the perturbed initial guess
    `missing`  : Bernoulli probability that a whole (camera, frame) detection is NaN.
    `outlier_frames` : that many frames get a grossly wrong initial pose (exercise the pre-filter).
    `scalar_nans` : that many single (u or v) scalars are set to NaN (per-coordinate masking).
    `frame_seed` : if given, board poses / noise / pose perturbations come from this seed while the cameras
                   still come from `seed` and `perturb_seed` -- frame shards of ONE rig for multi-GPU runs."""
    rng = np.random.default_rng(seed)
    obj = board_points(rows, cols, pitch)
    N = obj.shape[0]
    C, F = n_cameras, n_frames
    centre = obj.mean(0)

    # cameras on a ring, looking at the world origin
    phi = 2 * np.pi * (np.arange(C) + 0.25 * rng.uniform(-1, 1, C)) / C
    pos = np.stack([600 * np.cos(phi), 600 * np.sin(phi), np.full(C, 400.0)], -1)
    ext = np.zeros((C, 6))
    for c in range(C):
        z = -pos[c] / np.linalg.norm(pos[c])
        xax = np.cross(z, np.array([0.0, 0.0, 1.0]))
        xax /= np.linalg.norm(xax)
        R = np.stack([xax, np.cross(z, xax), z])
        ext[c, :3] = _rotvec(R)
        ext[c, 3:] = -R @ pos[c]
    cam = np.zeros((C, 12))
    cam[:, 0] = rng.uniform(1100, 1200, C)
    cam[:, 1] = cam[:, 0] * rng.uniform(0.99, 1.01, C)
    cam[:, 2] = 640 + rng.uniform(-15, 15, C)
    cam[:, 3] = 512 + rng.uniform(-15, 15, C)
    cam[:, 4] = rng.uniform(-0.1, -0.05, C)
    cam[:, 5] = 0.02 + rng.uniform(-0.005, 0.005, C)

    # board poses: rotate about the board centre, then scatter around the origin
    if frame_seed is not None:
        rng = np.random.default_rng([int(frame_seed), 0x6672616D])
    rv = rng.normal(0, 0.6, (F, 3))
    tr = rng.normal(0, 60.0, (F, 3))
    poses = np.concatenate([rv, tr - np.einsum("fij,j->fi", _rot(rv), centre)], -1)

    # world := camera 0 frame  (camera 0 extrinsic becomes the exact zero vector)
    T0 = _T(ext[0])
    Tc = _T(ext) @ np.linalg.inv(T0)
    ext = _t6(Tc)
    ext[0] = 0.0
    poses = _t6(T0 @ _T(poses))
    cam[:, 6:] = ext

    uvs = project(cam, poses, obj) + rng.normal(0, noise, (C, F, N, 2))
    if missing > 0:
        gone = rng.uniform(size=(C, F)) < missing
        uvs[gone] = np.nan
    if scalar_nans:
        idx = rng.choice(uvs.size, scalar_nans, replace=False)
        uvs.reshape(-1)[idx] = np.nan

    prng = np.random.default_rng(perturb_seed)
    cam0 = cam.copy()
    cam0[:, 0:2] *= 1 + 0.01 * prng.normal(size=(C, 2))
    cam0[:, 2:4] += 3.0 * prng.normal(size=(C, 2))
    cam0[:, 4:6] *= 1 + 0.1 * prng.normal(size=(C, 2))
    cam0[1:, 6:9] += 1e-3 * prng.normal(size=(C - 1, 3))
    cam0[1:, 9:12] += 0.5 * prng.normal(size=(C - 1, 3))
    if frame_seed is not None:
        prng = np.random.default_rng([int(frame_seed), int(perturb_seed), 0x70657274])
    poses0 = poses.copy()
    poses0[:, :3] += 1e-3 * prng.normal(size=(F, 3))
    poses0[:, 3:] += 0.5 * prng.normal(size=(F, 3))
    if outlier_frames:
        bad = prng.choice(F, outlier_frames, replace=False)
        poses0[bad, 3:] += 40.0

    intr = []
    for c in range(C):
        K = np.eye(3)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = cam0[c, :4]
        intr.append((K, np.array([cam0[c, 4], cam0[c, 5], 0.0, 0.0, 0.0])))
    return dict(uvs=uvs, obj=obj, true_cam=cam, true_poses=poses, extrinsics=cam0[:, 6:].copy(), intrinsics=intr, poses=poses0)
