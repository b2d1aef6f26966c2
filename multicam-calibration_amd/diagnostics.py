"""Reprojection diagnostics and stand-alone undistortion on the GPU.

`reprojection_errors` is the numeric core of the reference's `plot_residuals` (multicam_calibration/viz.py:70-210, the
computation at :160-186) with the same inputs; it returns what that function returns besides the figure:
(median_error, reprojections, transformed_reprojections).  Plotting itself stays out of scope.
`undistort_points` mirrors geometry.py:328-358 (a wrapper around cv2.undistortPoints(uvs, K, dist, None, K)).

Both numerical kernels of the reference live in OpenCV (cv2.undistortPoints, cv2.findHomography + cv2.perspectiveTransform),
which is absent here: parity with cv2's numbers is UNPINNED; the GPU results are checked against a numpy restatement of the
published algorithms (oracle/diagnostics_oracle.py) -- fixed-point undistortion; homography = the least-squares minimiser of
the transfer error in the board plane, which is what findHomography(method=0) refines its normalised-DLT estimate to.
"""
import numpy as np

from . import ops
from .api import serialize_params


def _dist5(all_intrinsics):
    d5 = np.zeros((len(all_intrinsics), 5))
    for c, (_, d) in enumerate(all_intrinsics):
        d = np.ravel(np.asarray(d, dtype=np.float64))
        if d.size > 5 and np.any(d[5:] != 0):
            raise NotImplementedError("only the 5-coefficient distortion model (k1, k2, p1, p2, k3) is supported")
        d5[c, : min(5, d.size)] = d[:5]
    return d5


def undistort_points(uvs, camera_matrix, dist_coefs, device=0, iterations=5):
    """(...,2) pixel coordinates -> undistorted pixel coordinates for the same camera matrix; rows with a NaN stay NaN."""
    K = np.asarray(camera_matrix, dtype=np.float64)
    if K[0, 1] != 0:
        raise NotImplementedError("camera matrices with skew are not supported")
    uvs = np.asarray(uvs, dtype=np.float64)
    d5 = _dist5([(K, dist_coefs)])[0]
    return ops.undistort_points(uvs, np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), d5, iterations, device).reshape(uvs.shape)


def reprojection_errors(all_calib_uvs, all_extrinsics, all_intrinsics, calib_objpoints, calib_poses, device=0, undistort_iterations=5, arrays=True):
    """Per-camera median reprojection error in the board plane, as `plot_residuals` reports it (viz.py:160-186).

    For every camera the board points (posed by `calib_poses`) are projected WITHOUT distortion (viz.py:166-168); for
    every frame whose detection is complete in that camera a homography maps the undistorted detections to the board's XY
    (viz.py:169-173), the reprojections go through it (:174-176), and the error is their distance to the board points.
    Returns (median_error (C,), reprojections (C,F,N,2), transformed_reprojections (C,F,N,2) with NaN where the board was
    not completely seen); the arrays are None with arrays=False (only the medians leave the GPU)."""
    uvs = np.asarray(all_calib_uvs, dtype=np.float64)
    obj = np.asarray(calib_objpoints, dtype=np.float64)
    if np.ptp(obj[:, 2]) != 0:
        raise NotImplementedError("the board-plane homography needs a planar calibration object (z = const)")
    prob = ops.Problem(uvs, obj, device=device)
    try:
        prob.set_params(0, serialize_params(all_extrinsics, all_intrinsics, np.asarray(calib_poses, dtype=np.float64)))
        return prob.reprojection_diagnostics(0, _dist5(all_intrinsics), undistort_iterations, arrays)
    finally:
        prob.close()
