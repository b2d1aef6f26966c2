"""GPU pre-filter, frame subsets, undistortion and reprojection diagnostics (csrc/mcba_diag.hip) against numpy / the oracles."""
import warnings

import numpy as np
import pytest

from oracle import ba_oracle as orc
from oracle import diagnostics_oracle as dgo
from oracle import triangulate_oracle as tri

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import multicam_calibration_amd as m

    m.ops.load_library()
    return m


def test_frame_errors_and_exact_median_vs_numpy(mc):
    """k_frame_err + radix select = the reference's pre-filter statistics (bundle_adjustment.py:265-282)."""
    p = mc.synth.make_problem(4, 333, seed=51, missing=0.3, scalar_nans=40, outlier_frames=9)   # ragged frame block, incomplete detections
    C, F, N = p["uvs"].shape[:3]
    x = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = mc.ops.Problem(p["uvs"], p["obj"])
    prob.set_params(0, x)
    mean_cf, full_cf = prob.frame_errors(0)
    err = np.linalg.norm(p["uvs"] - orc.predict_from_x(x, C, p["obj"]), axis=-1)             # (C,F,N), NaN where a coordinate is missing
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        want_mean = np.nanmean(err, axis=-1)
    np.testing.assert_allclose(mean_cf, want_mean, rtol=1e-12, atol=0, equal_nan=True)
    np.testing.assert_array_equal(full_cf, (~np.isnan(p["uvs"]).any(-1)).sum(-1))
    med, cnt = prob.error_median(None)
    assert cnt == int((~np.isnan(err)).sum())
    assert abs(med - np.nanmedian(err)) <= 1e-13 * med
    for k in (0, 1):   # an odd and an even number of values: the two-middle-values rule of np.median
        mask = np.zeros(F, np.uint8)
        mask[k:200:3] = 1
        med, cnt = prob.error_median(mask)
        sel = err[:, mask.astype(bool)]
        assert cnt == int((~np.isnan(sel)).sum())
        assert abs(med - np.nanmedian(sel)) <= 1e-13 * med
    mask[:] = 0
    med, cnt = prob.error_median(mask)
    assert cnt == 0 and np.isnan(med)                                                           # np.nanmedian of nothing
    prob.close()


def test_subset_handle_equals_a_fresh_upload(mc):
    p = mc.synth.make_problem(3, 150, seed=52, missing=0.2)
    x = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    frames = np.random.default_rng(1).permutation(150)[:77]
    full = mc.ops.Problem(p["uvs"], p["obj"], loss="cauchy", f_scale=0.7)
    sub = full.subset(frames, loss="cauchy", f_scale=0.7)
    full.close()                                                                                # the subset owns its own buffers
    ref = mc.ops.Problem(p["uvs"][:, frames], p["obj"], loss="cauchy", f_scale=0.7)
    xs = np.concatenate([x[:36], x[36:].reshape(150, 6)[frames].ravel()])
    for pr in (sub, ref):
        pr.set_params(0, xs)
        pr.linearize(0)
        pr.build_reduced(1e-3)
    a, b = sub.get_reduced(), ref.get_reduced()
    for k in ("S0", "rhs", "gc", "diagU", "scal"):
        np.testing.assert_array_equal(a[k], b[k])                                               # same data, same kernels: bit-identical
    np.testing.assert_array_equal(sub.residuals(0), ref.residuals(0))
    sub.close(), ref.close()


def test_undistort_points_vs_oracle(mc):
    rng = np.random.default_rng(5)
    K = np.array([[1150.0, 0, 655.0], [0, 1140.0, 500.0], [0, 0, 1]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.01])
    uv = rng.uniform(0, 1280, (7, 33, 2))
    uv[2, 5] = np.nan
    uv[4, 9, 1] = np.nan
    got = mc.undistort_points(uv, K, dist)
    want = tri.undistort_points(uv, K, dist)
    assert got.shape == uv.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10, equal_nan=True)
    assert np.isnan(got[4, 9]).all() and np.isnan(got[2, 5]).all()
    np.testing.assert_allclose(mc.undistort_points(uv[0], K, np.zeros(5)), uv[0], atol=1e-10)  # no distortion: identity


def test_reprojection_errors_vs_oracle(mc):
    """The numeric core of plot_residuals (viz.py:160-186): medians, reprojections and board-plane coordinates."""
    p = mc.synth.make_problem(3, 70, seed=53, missing=0.25, scalar_nans=9)                    # 70 frames: a full and a ragged wavefront
    intr = [(K, np.array([d[0], d[1], 4e-4, -3e-4, 5e-3])) for K, d in p["intrinsics"]]           # all five coefficients in play
    med, rep, tra = mc.reprojection_errors(p["uvs"], p["extrinsics"], intr, p["obj"], p["poses"])
    med0, rep0, tra0 = dgo.reprojection_errors(p["uvs"], p["extrinsics"], intr, p["obj"], p["poses"])
    np.testing.assert_allclose(rep, rep0, rtol=0, atol=1e-9)
    assert np.array_equal(np.isnan(tra), np.isnan(tra0))
    np.testing.assert_allclose(tra, tra0, rtol=0, atol=1e-6, equal_nan=True)                      # mm, on a 12.5 mm pitch: both sides
    np.testing.assert_allclose(med, med0, rtol=1e-7)                                              # stop at a flat transfer error
    med2, rep2, tra2 = mc.reprojection_errors(p["uvs"], p["extrinsics"], intr, p["obj"], p["poses"], arrays=False)
    assert rep2 is None and tra2 is None
    np.testing.assert_array_equal(med2, med)
    # a camera that never sees the whole board: np.median of nothing
    uvs = p["uvs"].copy()
    uvs[1, :, 0, 0] = np.nan
    med3 = mc.reprojection_errors(uvs, p["extrinsics"], intr, p["obj"], p["poses"], arrays=False)[0]
    assert np.isnan(med3[1]) and abs(med3[0] - med[0]) <= 1e-12 * med[0]


def test_noise_free_calibration_has_zero_board_plane_error(mc):
    p = mc.synth.make_problem(4, 40, seed=54, noise=0.0, missing=0.2)
    intr = [(np.array([[c[0], 0, c[2]], [0, c[1], c[3]], [0, 0, 1.0]]), np.array([c[4], c[5], 0, 0, 0])) for c in p["true_cam"]]
    med, rep, tra = mc.reprojection_errors(p["uvs"], p["true_cam"][:, 6:], intr, p["obj"], p["true_poses"])
    assert np.all(med < 1e-8)
