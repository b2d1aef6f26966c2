"""CPU oracle for `triangulate()` (reference geometry.py:361-433) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (`multicam-calibration_amd/`) never does.

PARITY UNPINNED.  The reference computes the two numerical kernels of this path inside OpenCV, which is absent from
this image and from /root/reference (dependency `opencv-python`, unpinned in setup.cfg:14-23):
  * cv2.undistortPoints(src, K, dist, None, K)  -- geometry.py:355-357.  Published algorithm (OpenCV
    modules/calib3d/src/undistort.dispatch.cpp, cvUndistortPointsInternal): normalise with K, then a fixed-point
    iteration  x <- (x0 - dx_tangential(x)) * icdist(x),  icdist = 1 / (1 + k1 r2 + k2 r4 + k3 r6)  for the 5-coefficient
    model, 5 iterations with the default termination criteria, and re-project with P = K.
  * cv2.triangulatePoints(P1, P2, x1, x2)       -- geometry.py:409-414.  Published algorithm (modules/calib3d/src/
    triangulate.cpp, icvTriangulatePoints): per point the 4x4 matrix with rows  x_j P_j[2] - P_j[0],  y_j P_j[2] - P_j[1]
    (j = 1, 2), SVD, the right singular vector of the smallest singular value is the homogeneous point.
Everything around them (pairing, NaN handling, nan-median over camera pairs, de-homogenisation: geometry.py:392-433,
254-274) is restated from the reference's own source.  With no cv2 to generate vectors the restatement is anchored on
the reference's call sites above and on exact recovery of synthetic truth (tests/test_triangulate_cpu.py).
"""
import numpy as np


def rodrigues(r):  # geometry.py:8-35
    r = np.asarray(r, dtype=np.float64)
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0.0]]) / (th if th > 0 else 1.0)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def projection_matrix(extrinsics, camera_matrix):  # geometry.py:199-229: P = K [R | t]
    ext = np.asarray(extrinsics, dtype=np.float64)
    return np.asarray(camera_matrix, dtype=np.float64) @ np.c_[rodrigues(ext[:3]), ext[3:]]


def undistort_points(uvs, camera_matrix, dist_coefs, iterations=5):
    """(...,2) pixels -> undistorted pixels (same K), NaN rows stay NaN (geometry.py:328-359)."""
    uvs = np.asarray(uvs, dtype=np.float64)
    K = np.asarray(camera_matrix, dtype=np.float64)
    k = np.zeros(5)
    k[: np.size(dist_coefs)] = np.ravel(dist_coefs)[:5]
    k1, k2, p1, p2, k3 = k
    x0 = (uvs[..., 0] - K[0, 2]) / K[0, 0]
    y0 = (uvs[..., 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iterations):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    out = np.stack([x * K[0, 0] + K[0, 2], y * K[1, 1] + K[1, 2]], axis=-1)
    out[np.isnan(uvs).any(-1)] = np.nan
    return out


def triangulate_pair(P1, P2, uv1, uv2):
    """Linear (DLT) triangulation of n points seen by two cameras -> (n,3)."""
    n = len(uv1)
    A = np.empty((n, 4, 4))
    A[:, 0] = uv1[:, 0:1] * P1[2] - P1[0]
    A[:, 1] = uv1[:, 1:2] * P1[2] - P1[1]
    A[:, 2] = uv2[:, 0:1] * P2[2] - P2[0]
    A[:, 3] = uv2[:, 1:2] * P2[2] - P2[1]
    X = np.linalg.svd(A)[2][:, 3]
    return X[:, :3] / X[:, 3:]


def triangulate(all_uvs, all_extrinsics, all_intrinsics, iterations=5):
    """Median over all camera pairs of the pairwise DLT triangulations (geometry.py:361-433)."""
    C = len(all_extrinsics)
    n = np.asarray(all_uvs[0]).shape[0]
    und = [undistort_points(uv, K, d, iterations) for uv, (K, d) in zip(all_uvs, all_intrinsics)]
    Ps = [projection_matrix(e, K) for e, (K, _) in zip(all_extrinsics, all_intrinsics)]
    pairwise = []
    for i in range(C):
        for j in range(i + 1, C):
            pts = np.full((n, 3), np.nan)
            seen = ~(np.isnan(und[i]).any(-1) | np.isnan(und[j]).any(-1))
            if seen.any():
                pts[seen] = triangulate_pair(Ps[i], Ps[j], und[i][seen], und[j][seen])
            pairwise.append(pts)
    pairwise = np.stack(pairwise)
    out = np.full((n, 3), np.nan)
    some = ~np.isnan(pairwise).all((0, 2))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        out[some] = np.nanmedian(pairwise[:, some], axis=0)
    return out
