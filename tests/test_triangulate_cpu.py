"""triangulate() oracle (oracle/triangulate_oracle.py, parity unpinned: see its header) against synthetic truth, and the
reference's NaN / pairing semantics (geometry.py:392-433) restated from its source."""
import numpy as np

import multicam_calibration_amd as m
from oracle import triangulate_oracle as tri


def scene(C=5, P=300, seed=0, noise=0.0, p_unseen=0.0):
    p = m.synth.make_problem(C, 2, seed=seed, noise=0.0)
    cam = p["true_cam"].copy()
    rng = np.random.default_rng(seed + 7)
    T = m.synth._T(p["true_poses"][0])
    X = rng.normal(0, 60, (P, 3)) @ T[:3, :3].T + T[:3, 3]   # a cloud where the board would be (world = camera 0)
    uvs = np.stack([m.synth.project(cam[c:c + 1], np.zeros((1, 6)), X)[0, 0] for c in range(C)])
    uvs += rng.normal(0, noise, uvs.shape) if noise else 0.0
    if p_unseen:
        uvs[rng.uniform(size=(C, P)) < p_unseen] = np.nan
    intr = [(np.array([[c[0], 0, c[2]], [0, c[1], c[3]], [0, 0, 1.0]]), np.r_[c[4:6], 0, 0, 0]) for c in cam]
    return list(uvs), cam[:, 6:], intr, X


def test_oracle_recovers_noise_free_points():
    uvs, ext, intr, X = scene()
    assert np.abs(tri.triangulate(uvs, ext, intr, iterations=20) - X).max() < 1e-9
    assert np.abs(tri.triangulate(uvs, ext, intr) - X).max() < 1e-6   # cv2's 5 undistortion iterations


def test_oracle_undistort_inverts_the_projection_model():
    uvs, ext, intr, X = scene(C=2, P=50)
    K, d = intr[1]
    cam_nodist = np.r_[K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0, 0, ext[1]]
    ideal = m.synth.project(cam_nodist[None], np.zeros((1, 6)), X)[0, 0]
    np.testing.assert_allclose(tri.undistort_points(uvs[1], K, d, iterations=25), ideal, atol=1e-9)


def test_oracle_nan_semantics():
    uvs, ext, intr, X = scene(C=4, P=40, p_unseen=0.45, seed=3)
    out = tri.triangulate(uvs, ext, intr, iterations=20)
    seen = (~np.isnan(np.stack(uvs)).any(-1)).sum(0)
    assert np.array_equal(np.isnan(out).any(1), seen < 2)          # fewer than two views -> NaN row (geometry.py:427-428)
    assert np.abs(out[seen >= 2] - X[seen >= 2]).max() < 1e-8
    uvs[2][5, 0] = np.nan                                            # one coordinate missing = camera does not see the point
    assert not np.isnan(tri.triangulate(uvs, ext, intr)[5]).any() or seen[5] < 3
