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
