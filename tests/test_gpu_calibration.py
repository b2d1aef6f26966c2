"""calibrate() without OpenCV on the GPU: the two OpenCV-backed steps are served by the library's
own LM (single-camera bundle adjustment with a parameter mask).  cv2 is absent here, so these tests pin behaviour to
the synthetic truth and to the property that matters downstream: bundle_adjust started from calibrate() ends in the
same optimum as from any other start."""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import multicam_calibration_amd as m

    m.ops.load_library()
    return m


def _pose_err(a, b):
    from multicam_calibration_amd import calibration as cal

    Ta, Tb = cal.get_transformation_matrix(a), cal.get_transformation_matrix(b)
    D = np.linalg.inv(Tb) @ Ta   # (rotation angle ~ |R - I|_F / sqrt 2; the reference's rodrigues_inv has no clip and
    return np.linalg.norm(D[..., :3, :3] - np.eye(3), axis=(-2, -1)).max() / np.sqrt(2), np.abs(D[..., :3, 3]).max()   # returns NaN at exactly I)


def test_get_intrinsics_and_estimate_pose_recover_truth(mc):
    p = mc.synth.make_problem(3, 80, seed=40, noise=0.1, missing=0.2)
    cam_true = p["true_cam"]
    np.random.seed(5)
    K, dist = mc.get_intrinsics(p["uvs"][1], p["obj"], (1280, 1024), n_samples=40)
    assert dist.shape == (5,) and np.all(dist[2:] == 0)
    # 40 views, 0.1 px noise: focal lengths / principal point to a few 1e-4 relative, k1 to a few percent
    np.testing.assert_allclose([K[0, 0], K[1, 1]], cam_true[1, :2], rtol=2e-3)
    np.testing.assert_allclose([K[0, 2], K[1, 2]], cam_true[1, 2:4], atol=3.0)
    assert abs(dist[0] - cam_true[1, 4]) < 0.02 and abs(dist[1] - cam_true[1, 5]) < 0.1
    assert K[0, 1] == 0 and K[1, 0] == 0 and K[2, 2] == 1

    # poses with the TRUE intrinsics: board -> camera 1, NaN exactly where the detection is incomplete
    Kt = np.array([[cam_true[1, 0], 0, cam_true[1, 2]], [0, cam_true[1, 1], cam_true[1, 3]], [0, 0, 1.0]])
    poses = mc.estimate_pose(p["uvs"][1], p["obj"], Kt, np.r_[cam_true[1, 4:6], 0, 0, 0])
    missing = np.isnan(p["uvs"][1]).any((1, 2))
    assert np.array_equal(np.isnan(poses).any(1), missing) and (~missing).sum() > 40
    from multicam_calibration_amd import calibration as cal

    want = cal.get_transformation_vector(cal.get_transformation_matrix(cam_true[1, 6:])[None] @ cal.get_transformation_matrix(p["true_poses"]))
    er, et = _pose_err(poses[~missing], want[~missing])
    assert er < 2e-2 and et < 3.0   # rad / mm at 0.1 px noise: a 100 mm board 700 mm away constrains its tilt only weakly

    # noise-free detections: the minimiser IS the truth
    q = mc.synth.make_problem(3, 80, seed=40, noise=0.0, missing=0.2)
    poses = mc.estimate_pose(q["uvs"][1], q["obj"], Kt, np.r_[cam_true[1, 4:6], 0, 0, 0])
    er, et = _pose_err(poses[~missing], want[~missing])
    assert er < 1e-8 and et < 1e-6
    np.random.seed(5)
    K0, dist0 = mc.get_intrinsics(q["uvs"][1], q["obj"], (1280, 1024), n_samples=40)
    np.testing.assert_allclose([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2]], cam_true[1, :4], rtol=1e-8)
    np.testing.assert_allclose(dist0[:2], cam_true[1, 4:6], rtol=1e-6)


def test_same_random_draw_as_the_reference(mc):
    """get_intrinsics consumes the global numpy RNG exactly like calibration.py:57-60 (one choice without replacement
    over the complete detections)."""
    p = mc.synth.make_problem(2, 30, seed=41, missing=0.3)
    np.random.seed(11)
    mc.get_intrinsics(p["uvs"][0], p["obj"], (1280, 1024), n_samples=10)
    after = np.random.randint(1 << 30)
    np.random.seed(11)
    n_complete = int((~np.isnan(p["uvs"][0]).any((1, 2))).sum())
    np.random.choice(n_complete, 10, replace=False)
    assert after == np.random.randint(1 << 30)


def test_calibrate_then_bundle_adjust_reaches_the_same_optimum(mc, capsys):
    p = mc.synth.make_problem(4, 120, seed=42, noise=0.2, missing=0.25)
    np.random.seed(3)
    ext, intr, poses, tree = mc.calibrate(p["uvs"], [(1280, 1024)] * 4, p["obj"], root=0, verbose=True, n_samples_for_intrinsics=50)
    out = capsys.readouterr().out
    for line in ("Estimating camera intrinsics", "Initializing calibration object poses", "Estimating camera extrinsics", "Merging calibration object poses"):
        assert line in out
    assert ext.shape == (4, 6) and np.all(ext[0] == 0) and len(intr) == 4 and poses.shape == (120, 6) and len(tree) == 3
    seen = ~np.isnan(p["uvs"]).any((2, 3))
    assert np.array_equal(np.isnan(poses).any(1), ~seen.any(0))
    # initial extrinsics close to the truth (world = camera 0 in both)
    er, et = _pose_err(ext[1:], p["true_cam"][1:, 6:])
    assert er < 3e-2 and et < 10.0   # an initialisation (per-camera intrinsics differ at 0.2 px noise), not the answer

    with contextlib.redirect_stdout(io.StringIO()):
        a = mc.bundle_adjust(p["uvs"], ext, intr, p["obj"], poses, n_frames=None, ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=200, verbose=0, return_jac=False)
        b = mc.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=200, verbose=0,
                             return_jac=False)
    np.testing.assert_array_equal(a[3], b[3])
    assert abs(a[4].cost - b[4].cost) <= 1e-9 * b[4].cost
    ca, cb = a[4].x[:48].reshape(4, 12), b[4].x[:48].reshape(4, 12)
    assert (np.abs(ca[:, :6] - cb[:, :6]) / np.abs(cb[:, :6])).max() < 1e-6   # intrinsics + distortion are gauge-free
