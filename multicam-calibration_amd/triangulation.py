"""`triangulate()` on the GPU (reference geometry.py:328-433).

Same signature and return value as the reference.  The reference undistorts with cv2.undistortPoints, triangulates every
camera pair with cv2.triangulatePoints and takes the nan-median over the pairs; here one HIP kernel does all three per
point (`csrc/mcba_triangulate.hip`: OpenCV's fixed-point undistortion for the 5-coefficient model, the 4x4 DLT null vector
by one-sided Jacobi, a sorting network for the median; beyond 8 cameras one wavefront per point with the camera pairs
across its lanes and the median by rank counting).  OpenCV is absent from this image, so parity with cv2's numbers is
unpinned; the kernel is checked against a numpy restatement of the two published algorithms (oracle/triangulate_oracle.py)
and against exact recovery of synthetic points.
"""
import ctypes

import numpy as np

from . import ops


def _cam_blocks(all_extrinsics, all_intrinsics):
    C = len(all_extrinsics)
    cam = np.zeros((C, 12))
    dist = np.zeros((C, 5))
    for c, (ext, (K, d)) in enumerate(zip(all_extrinsics, all_intrinsics)):
        K = np.asarray(K, dtype=np.float64)
        if K[0, 1] != 0:
            raise NotImplementedError("camera matrices with skew are not supported")
        d = np.ravel(np.asarray(d, dtype=np.float64))
        if d.size > 5 and np.any(d[5:] != 0):
            raise NotImplementedError("only the 5-coefficient distortion model (k1, k2, p1, p2, k3) is supported")
        cam[c, :4] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        cam[c, 6:] = np.asarray(ext, dtype=np.float64)
        dist[c, : min(5, d.size)] = d[:5]
    cam[:, 4:6] = dist[:, :2]
    return cam, dist


def triangulate(all_uvs, all_extrinsics, all_intrinsics, device=0, undistort_iterations=5, return_kernel_ms=False):
    """all_uvs: per camera (n_points, 2), NaN = not seen.  Returns (n_points, 3); NaN rows where fewer than two cameras see
    the point (geometry.py:361-433)."""
    lib = ops.load_library()
    uvs = np.ascontiguousarray(np.stack([np.asarray(u, dtype=np.float64) for u in all_uvs]))
    if uvs.ndim != 3 or uvs.shape[2] != 2 or uvs.shape[0] != len(all_extrinsics) or len(all_extrinsics) != len(all_intrinsics):
        raise ValueError("all_uvs must be one (n_points, 2) array per camera, matching all_extrinsics / all_intrinsics")
    C, P = uvs.shape[:2]
    if not 2 <= C <= 64:
        raise NotImplementedError("triangulate() supports 2 to 64 cameras")
    cam, dist = _cam_blocks(all_extrinsics, all_intrinsics)
    out = np.empty((P, 3))
    ms = ctypes.c_double(0.0)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.mcba_triangulate(C, P, uvs.ctypes.data_as(dp), cam.ctypes.data_as(dp), dist.ctypes.data_as(dp), int(undistort_iterations), int(device),
                              out.ctypes.data_as(dp), ctypes.cast(ctypes.byref(ms), dp))
    if rc != ops.OK:
        raise ops.McbaError(rc, lib.mcba_last_error().decode())
    return (out, ms.value) if return_kernel_ms else out
