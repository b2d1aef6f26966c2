"""CPU oracle for the reprojection diagnostics (reference viz.py:160-186) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product never does.

PARITY UNPINNED: the reference computes this path with three OpenCV calls (dependency `opencv-python`, unpinned in
setup.cfg:14-23, absent from this image and from /root/reference):
  * cv2.undistortPoints(uvs, K, dist, None, K)      geometry.py:355-357   -> triangulate_oracle.undistort_points
  * cv2.findHomography(src, dst)  (method 0)        viz.py:173.  Published algorithm (OpenCV modules/calib3d/src/
    fundam.cpp): Hartley normalisation of both point sets, the homogeneous linear estimate (smallest eigenvector of A^T A),
    then a Levenberg-Marquardt refinement of the transfer error sum |dst - H src|^2 over the 8 parameters with h33 = 1.
    Restated here as: normalised linear estimate, then Levenberg-Marquardt on the same transfer error run to convergence --
    the least-squares homography a converged refinement ends in.
  * cv2.perspectiveTransform(points, H)             viz.py:174-176: (H [x y 1])_xy / (H [x y 1])_z
The surrounding logic (distortion-free projection, completeness test, error norm, median: viz.py:160-186) is restated from
the reference's own source.  Anchors: those call sites, and exact recovery on noise-free synthetic data in the tests.
"""
import numpy as np

from oracle import ba_oracle as orc
from oracle import triangulate_oracle as tri


def find_homography(src, dst, lm_iterations=60):
    """Least-squares homography dst ~ H src, (n,2) each: Hartley normalisation, linear start (the better of the homogeneous
    SVD estimate and the h33 = 1 normal-equation estimate), Levenberg-Marquardt on the transfer error."""
    def norm(P):
        c = P.mean(0)
        s = np.sqrt(2.0) / np.mean(np.linalg.norm(P - c, axis=1))
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (P - c) * s, T

    sn, Ts = norm(src)
    dn, Td = norm(dst)
    x, y, X, Y = sn[:, 0], sn[:, 1], dn[:, 0], dn[:, 1]
    z, o = np.zeros_like(x), np.ones_like(x)
    A = np.concatenate([np.stack([x, y, o, z, z, z, -X * x, -X * y], 1), np.stack([z, z, z, x, y, o, -Y * x, -Y * y], 1)])
    b = np.concatenate([X, Y])

    def model(h):
        w = h[6] * x + h[7] * y + 1
        return (h[0] * x + h[1] * y + h[2]) / w, (h[3] * x + h[4] * y + h[5]) / w, w

    def error(h):
        px, py, _ = model(h)
        return np.sum((X - px) ** 2 + (Y - py) ** 2)

    starts = [np.linalg.lstsq(A, b, rcond=None)[0]]
    hs = np.linalg.svd(np.concatenate([A, -b[:, None]], 1))[2][-1]
    if abs(hs[8]) > 1e-12:
        starts.append(hs[:8] / hs[8])
    h = min(starts, key=error)
    e_cur, mu = error(h), 1e-4
    for _ in range(lm_iterations):
        px, py, w = model(h)
        J = np.concatenate([np.stack([x / w, y / w, 1 / w, z, z, z, -px * x / w, -px * y / w], 1), np.stack([z, z, z, x / w, y / w, 1 / w, -py * x / w, -py * y / w], 1)])
        r = np.concatenate([X - px, Y - py])
        M = J.T @ J
        M[np.diag_indices(8)] *= 1 + mu
        step = np.linalg.solve(M, J.T @ r)
        e_new = error(h + step)
        if e_new <= e_cur:
            h, e_cur, mu = h + step, e_new, max(mu * 0.1, 1e-15)
        else:
            mu = min(mu * 10, 1e8)
    Hn = np.append(h, 1.0).reshape(3, 3)
    H = np.linalg.inv(Td) @ Hn @ Ts
    return H / H[2, 2]


def perspective_transform(pts, H):
    q = np.c_[pts, np.ones(len(pts))] @ H.T
    return q[:, :2] / q[:, 2:]


def reprojection_errors(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, iterations=5):
    """(median_error, reprojections, transformed_reprojections) of plot_residuals (viz.py:160-186)."""
    C, F, N, _ = all_calib_uvs.shape
    median_error = np.zeros(C)
    reprojections = np.zeros((C, F, N, 2))
    transformed = np.zeros((C, F, N, 2)) * np.nan
    pts = orc.embed_calib_objpoints(calib_objpoints, calib_poses)
    for cam in range(C):
        K, dist = all_intrinsics[cam]
        reprojections[cam] = orc.project_points(pts, np.asarray(all_extrinsics[cam]), K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.0, 0.0)
        und = tri.undistort_points(all_calib_uvs[cam], K, dist, iterations)
        valid = np.nonzero(~np.isnan(und).any((-1, -2)))[0]
        for t in valid:
            H = find_homography(und[t], calib_objpoints[:, :2])
            transformed[cam, t] = perspective_transform(reprojections[cam, t], H)
        errors = np.linalg.norm(transformed[cam, valid] - calib_objpoints[:, :2], axis=-1)
        median_error[cam] = np.median(errors) if errors.size else np.nan
    return median_error, reprojections, transformed
