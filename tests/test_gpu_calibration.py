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


# ---------------------------------------------------------------------------------------------------------------------------------
# the five-coefficient model: get_intrinsics(fix_k3=False | zero_tangent_dist=False) (calibration.py:11-71 -> cv2.calibrateCamera without the flags)
def _views(n_views, seed, intr9, noise=0.0, rows=6, cols=9, pitch=25.0):
    """Synthetic views of a planar board through oracle/calibration_oracle.py's restatement of OpenCV's published model."""
    from oracle import calibration_oracle as co

    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(cols) - (cols - 1) / 2, np.arange(rows) - (rows - 1) / 2)
    obj = np.stack([gx.ravel() * pitch, gy.ravel() * pitch, np.zeros(rows * cols)], -1)
    poses = np.concatenate([rng.normal(0, 0.35, (n_views, 3)), rng.normal(0, 60, (n_views, 2)), rng.uniform(450, 800, (n_views, 1))], axis=1)
    poses[0, :3] = 0.0            # a board exactly facing the camera (rotation vector 0: the theta = 0 convention)
    poses[1, :3] = [1e-5, -2e-5, 0.0]   # ... and one in the series branch
    uvs = np.stack([co.project5(obj, ps, intr9) for ps in poses])
    return obj, poses, uvs + rng.normal(0, noise, uvs.shape)


TRUE9 = np.array([1210.0, 1195.0, 655.0, 500.0, -0.21, 0.09, 1.5e-3, -8e-4, -0.03])


def test_calibration_normal_equations_vs_oracle(mc):
    """The GPU's per-view Gauss-Newton blocks / gradients / costs (forward-mode automatic differentiation) against the oracle's, built from a
    3-point finite-difference Jacobian of its numpy restatement of the model: 15 parameters per view, incl. a view at rotation vector 0."""
    from oracle import calibration_oracle as co

    obj, poses, uvs = _views(7, 3, TRUE9, noise=0.4)
    rng = np.random.default_rng(4)
    k = TRUE9 * (1 + rng.normal(0, 0.02, 9))
    ps = poses + rng.normal(0, 1e-2, poses.shape) * np.r_[1, 1, 1, 50, 50, 50]
    ps[0, :3] = 0.0
    uvs[3, 5, 1] = np.nan   # a missing scalar contributes nothing
    H, g, c = mc.ops.calib_normal_equations(uvs, obj, k, ps)
    for v in range(len(ps)):
        uv = uvs[v].copy()
        seen = ~np.isnan(uv)

        def res(p, uv=uv, seen=seen):
            return np.where(seen, np.nan_to_num(uv) - co.project5(obj, p[9:], p[:9]), 0.0).ravel()

        from scipy.optimize._numdiff import approx_derivative

        p15 = np.concatenate([k, ps[v]])
        J = approx_derivative(res, p15, method="3-point")
        r = res(p15)
        Ho, go, cost = J.T @ J, J.T @ r, 0.5 * r @ r
        sc = np.sqrt(np.outer(np.diag(Ho), np.diag(Ho)))
        assert (np.abs(H[v] - Ho) / sc).max() < 1e-6, v
        assert (np.abs(g[v] - go) / np.sqrt(np.diag(Ho) * 2 * cost)).max() < 1e-6, v
        assert abs(c[v] - cost) <= 1e-12 * cost, v


@pytest.mark.parametrize("fix_k3,zero_tangent", [(False, False), (True, False), (False, True)])
def test_get_intrinsics_five_coefficient_model(mc, fix_k3, zero_tangent):
    """Noise-free views of a camera whose distortion lies in the asked-for model: the minimiser is the truth; with noise: the optimum of
    calibrateCamera's objective as scipy finds it from the same closed-form start (oracle/calibration_oracle.py); the held coefficients stay 0."""
    from multicam_calibration_amd import calibration as cal
    from oracle import calibration_oracle as co

    true9 = TRUE9.copy()
    if fix_k3:
        true9[8] = 0.0
    if zero_tangent:
        true9[6:8] = 0.0
    obj, poses, uvs = _views(30, 11, true9)
    np.random.seed(2)
    K, dist = mc.get_intrinsics(uvs, obj, (1280, 1024), n_samples=30, fix_k3=fix_k3, zero_tangent_dist=zero_tangent)
    got = np.r_[K[0, 0], K[1, 1], K[0, 2], K[1, 2], dist]
    assert dist.shape == (5,) and K[0, 1] == 0 and K[2, 2] == 1
    np.testing.assert_allclose(got[:4], true9[:4], rtol=1e-7)
    np.testing.assert_allclose(got[4:], true9[4:], rtol=1e-5, atol=1e-9)
    if fix_k3:
        assert dist[4] == 0.0
    if zero_tangent:
        assert dist[2] == 0.0 and dist[3] == 0.0
    # the same draw from the global RNG as the reference (calibration.py:57-60), whatever the flags
    after = np.random.randint(1 << 30)
    np.random.seed(2)
    np.random.choice(30, 30, replace=False)
    assert after == np.random.randint(1 << 30)

    # with noise: scipy's optimum of the same objective from the same start
    obj, poses, uvs = _views(25, 12, true9, noise=0.3)
    np.random.seed(3)
    K, dist = mc.get_intrinsics(uvs, obj, (1280, 1024), n_samples=25, fix_k3=fix_k3, zero_tangent_dist=zero_tangent)
    np.random.seed(3)
    sel = uvs[np.random.choice(25, 25, replace=False)]
    K0, poses0 = cal._zhang_start(sel, obj, (1280, 1024))
    free9 = np.array([1, 1, 1, 1, 1, 1, not zero_tangent, not zero_tangent, not fix_k3], dtype=bool)
    k_o, ps_o, cost_o = co.refine(sel, obj, np.r_[K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], np.zeros(5)], poses0, free9)
    got = np.r_[K[0, 0], K[1, 1], K[0, 2], K[1, 2], dist]
    np.testing.assert_allclose(got[:4], k_o[:4], rtol=1e-6)
    np.testing.assert_allclose(got[4:], k_o[4:], rtol=1e-4, atol=1e-7)
    cost_here = sum(0.5 * np.sum((sel[v] - co.project5(obj, ps, got)) ** 2) for v, ps in enumerate(cal._refine_five_coefficients(sel, obj, got, ps_o, np.zeros(9, bool), 0)[1]))
    assert abs(cost_here - cost_o) <= 1e-8 * cost_o


def test_estimate_pose_with_five_coefficients(mc):
    """estimate_pose with tangential / k3 coefficients (what get_intrinsics returns with a flag off): NaN rows where the detection is incomplete,
    the truth from noise-free detections, scipy's per-view optimum with noise."""
    from oracle import calibration_oracle as co
    from scipy.optimize import least_squares

    obj, poses, uvs = _views(24, 21, TRUE9)
    uvs[5, 3] = np.nan
    K = np.array([[TRUE9[0], 0, TRUE9[2]], [0, TRUE9[1], TRUE9[3]], [0, 0, 1.0]])
    got = mc.estimate_pose(uvs, obj, K, TRUE9[4:])
    assert np.isnan(got[5]).all() and not np.isnan(np.delete(got, 5, 0)).any()
    er, et = _pose_err(np.delete(got, 5, 0)[2:], np.delete(poses, 5, 0)[2:])   # (views 0 / 1 sit at rotation 0: compared through their predictions below)
    assert er < 1e-8 and et < 1e-6
    for v in (0, 1):
        assert np.abs(co.project5(obj, got[v], TRUE9) - uvs[v]).max() < 1e-8
    obj, poses, uvs = _views(12, 22, TRUE9, noise=0.3)
    got = mc.estimate_pose(uvs, obj, K, TRUE9[4:])
    for v in range(12):
        ref = least_squares(lambda p: (uvs[v] - co.project5(obj, p, TRUE9)).ravel(), poses[v], xtol=1e-15, ftol=1e-15, gtol=1e-12)
        c_here = 0.5 * np.sum((uvs[v] - co.project5(obj, got[v], TRUE9)) ** 2)
        assert abs(c_here - ref.cost) <= 1e-9 * ref.cost, v
