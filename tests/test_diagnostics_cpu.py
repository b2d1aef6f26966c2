"""The diagnostics oracle (numpy restatement of viz.py:160-186 with OpenCV's published algorithms) anchored on exact recovery."""
import numpy as np

from oracle import ba_oracle as orc
from oracle import diagnostics_oracle as dgo
from oracle import triangulate_oracle as tri
from multicam_calibration_amd import synth


def _truth(p):
    intr = [(np.array([[c[0], 0, c[2]], [0, c[1], c[3]], [0, 0, 1.0]]), np.array([c[4], c[5], 0, 0, 0])) for c in p["true_cam"]]
    return p["true_cam"][:, 6:], intr


def test_homography_recovers_a_known_map():
    rng = np.random.default_rng(0)
    H = np.array([[1.1, 0.05, 3.0], [-0.02, 0.9, -2.0], [1e-4, -2e-4, 1.0]])
    src = rng.uniform(-100, 100, (30, 2))
    dst = dgo.perspective_transform(src, H)
    np.testing.assert_allclose(dgo.find_homography(src, dst), H, rtol=0, atol=1e-11)
    # with noise: a stationary point of the transfer error (what findHomography's refinement minimises)
    dstn = dst + rng.normal(0, 0.3, dst.shape)
    Hn = dgo.find_homography(src, dstn)
    base = np.sum((dgo.perspective_transform(src, Hn) - dstn) ** 2)
    for k in range(8):
        for eps in (1e-6, -1e-6):
            Hp = Hn.copy()
            Hp.flat[k] *= 1 + eps
            assert np.sum((dgo.perspective_transform(src, Hp) - dstn) ** 2) >= base * (1 - 1e-12)


def test_noise_free_calibration_has_zero_board_plane_error():
    p = synth.make_problem(3, 14, seed=3, missing=0.25, noise=0.0, scalar_nans=4)
    ext, intr = _truth(p)
    med, rep, tra = dgo.reprojection_errors(p["uvs"], ext, intr, p["obj"], p["true_poses"])
    assert np.all(med < 1e-9)                                        # mm: undistortion (5 rounds) + homography are exact here
    complete = ~np.isnan(p["uvs"]).any((-1, -2))
    assert np.array_equal(~np.isnan(tra).any((-1, -2)), complete)    # NaN exactly where the board was not completely seen
    np.testing.assert_allclose(tra[complete], np.broadcast_to(p["obj"][:, :2], tra[complete].shape), atol=1e-8)
    # the reprojections are distortion-free projections: undistorting the (noise-free) detections gives them back
    und = np.stack([tri.undistort_points(p["uvs"][c], *intr[c]) for c in range(3)])
    np.testing.assert_allclose(und[complete], rep[complete], atol=1e-8)
