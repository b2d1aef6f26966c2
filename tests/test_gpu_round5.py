"""GPU tests of the round-5 additions to the C ABI (ABI 6): the pre-filter in one crossing with the selection on the device
(mcba_prefilter: three-pass radix-select median, masks, exclusion), the device-resident LM loop in one crossing (mcba_lm_run) and the
packed solution + gradient (mcba_lm_result) -- each against the per-call sequence it replaces (which the earlier tests pin to the
oracle and to the reference's outputs), bit for bit, and against the oracle's pre-filter directly.  Run with `-m gpu` on an MI355X."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import ba_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    from conftest import GOLDEN

    return lambda name: np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="module")
def mc():
    import multicam_calibration_amd as m

    m.ops.load_library()
    return m


@contextlib.contextmanager
def env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def captured(f, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = f(*a, **k)
    return out, buf.getvalue()


def _select(mc, p, n_frames, thr, seed, **envkw):
    with env(**envkw):
        np.random.seed(seed)
        use, line = captured(mc.api.select_frames, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames, thr)
        return use, line, np.random.randint(0, 2**31 - 1)


CASES = [
    dict(C=3, F=70, kw=dict(seed=5, missing=0.2, scalar_nans=4, outlier_frames=3)),
    dict(C=6, F=333, kw=dict(seed=1, missing=0.3, outlier_frames=7)),
    dict(C=2, F=50, kw=dict(seed=0)),
    dict(C=4, F=64, kw=dict(seed=2, missing=0.6)),            # many frames seen by fewer than two cameras
    dict(C=9, F=129, kw=dict(seed=3, rows=2, cols=2, missing=0.1, outlier_frames=2)),
    dict(C=1, F=40, kw=dict(seed=4)),                         # one camera: no frame is complete in two
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fused_prefilter_equals_the_stepwise_one_and_the_oracle(mc, case):
    """mcba_prefilter against the per-call pre-filter (k_frame_err, host masks, eight-pass median: MCBA_PREFILTER_FUSED=0), against the same
    call with the eight-pass select forced behind the device masks (MCBA_PREFILTER_FALLBACK=1), and against the ORACLE's restatement of
    bundle_adjustment.py:265-296: same frames in the same order, same printed line to the last digit, same global-RNG state."""
    c = CASES[case]
    p = mc.synth.make_problem(c["C"], c["F"], **c["kw"])
    for n_frames, thr, seed in ((None, None, 0), (10, None, 1), (c["F"], None, 2), (7, 1.5, 3), (None, 0.25, 4), (None, float("nan"), 5), (None, 3, 6)):
        a = _select(mc, p, n_frames, thr, seed)
        b = _select(mc, p, n_frames, thr, seed, MCBA_PREFILTER_FUSED=0)
        f = _select(mc, p, n_frames, thr, seed, MCBA_PREFILTER_FALLBACK=1)
        np.testing.assert_array_equal(a[0], b[0])
        assert a[1] == b[1] and a[2] == b[2], (a[1], b[1])
        np.testing.assert_array_equal(a[0], f[0])
        assert a[1] == f[1] and a[2] == f[2], (a[1], f[1])
        np.random.seed(seed)
        import warnings
        with np.errstate(all="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            use_o, thr_o, _, line_o = orc.prefilter_frames(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames, thr)
        np.testing.assert_array_equal(a[0], use_o)
        assert np.random.randint(0, 2**31 - 1) == a[2]
        ta, to = a[1].strip().rsplit(" ", 1), line_o.rsplit(" ", 1)
        assert ta[0] == to[0]
        if not np.isnan(float(to[1])):
            assert abs(float(ta[1]) - float(to[1])) <= 1e-12 * abs(float(to[1]))   # (the oracle's errors differ from the kernel's in the last bits)


def test_prefilter_status_bits_and_info(mc):
    p = mc.synth.make_problem(5, 200, seed=9, missing=0.25, outlier_frames=5, scalar_nans=6)
    x = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = mc.ops.Problem(p["uvs"], p["obj"], upload=False)
    status, thr, info = prob.prefilter(x)
    mean_cf, full_cf = prob.frame_errors(0)          # the per-call kernels on the same handle
    complete = full_cf == p["uvs"].shape[2]
    used = complete.sum(0) > 1
    np.testing.assert_array_equal((status & 1).astype(bool), used)
    np.testing.assert_array_equal((status & 4).astype(bool), complete.all(0))
    mask = used.astype(np.uint8)
    med, cnt = prob.error_median(mask)
    assert thr == 5 * med and info[1] == med and info[2] == cnt and info[3] == 0
    worst = np.fmax.reduce(mean_cf, axis=0)
    excl = used & (np.nan_to_num(worst) > thr)
    np.testing.assert_array_equal((status & 2).astype(bool), excl)
    assert (info[4], info[5]) == (used.sum(), excl.sum())
    # a second call on the same handle (observations already there) with a caller's threshold
    status2, thr2, info2 = prob.prefilter(x, 0.4)
    assert thr2 == 0.4
    np.testing.assert_array_equal((status2 & 2).astype(bool), used & (np.nan_to_num(worst) > 0.4))
    prob.close()


def test_prefilter_median_at_scale_and_degenerate(mc):
    """The three-pass select at a size with millions of values (against numpy on the downloaded scores) and on data where every error is
    the SAME number (the candidate list overflows -> the eight-pass fallback must give the same answer)."""
    p = mc.synth.make_problem(6, 3000, seed=11, missing=0.1)
    x = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = mc.ops.Problem(p["uvs"], p["obj"], upload=False)
    status, thr, info = prob.prefilter(x)
    pred = orc.predict_from_x(x, 6, p["obj"])
    used = (status & 1).astype(bool)
    err = np.sqrt(((p["uvs"] - pred) ** 2).sum(-1))[:, used]
    v = np.sort(err[~np.isnan(err)])
    assert info[2] == v.size and info[3] == 0
    # the order statistics are exact: the device median lies between the oracle-side neighbours of the middle (its errors differ in the last bits)
    assert abs(info[1] - np.median(v)) <= 1e-10 * np.median(v)
    prob.close()
    # every observation exactly where the model predicts + a constant offset of (3, 4) px -> every error == 5 (to the bit or nearly): overflow path
    q = mc.synth.make_problem(2, 20000, seed=12, noise=0.0)
    xq = mc.serialize_params(q["extrinsics"], q["intrinsics"], q["poses"])
    uv = orc.predict_from_x(xq, 2, q["obj"]) + np.array([3.0, 4.0])
    prob = mc.ops.Problem(uv, q["obj"], upload=False)
    status, thr, info = prob.prefilter(xq)
    mean_cf, full_cf = prob.frame_errors(0)
    med, cnt = prob.error_median(np.ones(20000, np.uint8))
    assert info[1] == med and info[2] == cnt == 2 * 20000 * uv.shape[2] and abs(med - 5.0) < 1e-9
    assert info[3] == 1.0   # ~2 M values share the median's 24 leading bits: more than the candidate list holds
    prob.close()


def _ba(mc, p, **kw):
    np.random.seed(0)
    return captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)


@pytest.mark.parametrize("shape", [(2, 50, {}), (3, 70, dict(missing=0.2, scalar_nans=3)), (6, 400, dict(missing=0.1)), (12, 90, dict(rows=3, cols=4))])
def test_lm_run_equals_the_python_loop(mc, shape):
    """mcba_lm_run against the per-tick Python loop (MCBA_HOST_LOOP=1): the same ticks -- every trial cost, damping, ratio and step norm
    to the bit --, the same solution, counters, status, printed table; with limits (max_nfev) and both reduced-solver regimes."""
    C, F, kw = shape
    p = mc.synth.make_problem(C, F, seed=3, **kw)
    for opts in (dict(n_frames=None, verbose=2), dict(n_frames=None, verbose=0, ftol=1e-12, xtol=1e-12, gtol=1e-12), dict(n_frames=F // 2, verbose=1, max_nfev=4),
                 dict(n_frames=None, verbose=0, loss="cauchy", f_scale=2.0, max_nfev=30), dict(n_frames=None, verbose=0, fix_intrinsics=True), dict(n_frames=None, verbose=0, max_nfev=1)):
        (ea, ia, pa, ua, ra), outa = _ba(mc, p, **opts)
        with env(MCBA_HOST_LOOP=1):
            (eb, ib, pb, ub, rb), outb = _ba(mc, p, **opts)
        assert outa == outb
        np.testing.assert_array_equal(ua, ub)
        np.testing.assert_array_equal(ra.x, rb.x)
        np.testing.assert_array_equal(ra.grad, rb.grad)
        assert (ra.cost, ra.nfev, ra.njev, ra.status, ra.optimality) == (rb.cost, rb.nfev, rb.njev, rb.status, rb.optimality)
        assert ra.lm["history"] == rb.lm["history"] and ra.lm["iterations"] == rb.lm["iterations"] and ra.lm["lam"] == rb.lm["lam"]
        assert ra.lm["steps"] == rb.lm["steps"] and ra.lm["rebuilds"] == rb.lm["rebuilds"]
        np.testing.assert_array_equal(ra.fun, rb.fun)


def test_identity_selection_solves_on_the_prefilter_handle(mc):
    """n_frames=None with nothing excluded: no gather, the pre-filter's handle is the problem.  Same solution as the gathered path
    (forced through a random permutation of all frames: the minimiser does not depend on the frame order)."""
    p = mc.synth.make_problem(4, 150, seed=6)
    (e1, i1, p1, u1, r1), _ = _ba(mc, p, n_frames=None, verbose=0, ftol=1e-12, xtol=1e-12, gtol=1e-12)
    np.testing.assert_array_equal(u1, np.arange(150))
    (e2, i2, p2, u2, r2), _ = _ba(mc, p, n_frames=150, verbose=0, ftol=1e-12, xtol=1e-12, gtol=1e-12)   # np.random.choice permutes
    assert sorted(u2) == list(range(150)) and not np.array_equal(u2, u1)
    assert abs(r1.cost - r2.cost) <= 1e-9 * r1.cost
    np.testing.assert_allclose(p1[u2], p2, rtol=0, atol=1e-6)
    f = orc.residuals(r1.x, p["uvs"][:, u1], p["obj"])
    assert abs(orc.robust_cost(f) - r1.cost) <= 1e-10 * r1.cost


def test_lm_result_equals_the_separate_fetches(mc):
    for C, fixed in ((3, False), (3, True), (11, False)):
        p = mc.synth.make_problem(C, 80, seed=8, missing=0.15)
        x0 = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
        prob = mc.ops.Problem(p["uvs"], p["obj"])
        if fixed:
            assert prob.set_camera_block(6)
        lm = mc.solver.LevenbergMarquardt(prob, ftol=1e-10)
        lm.max_nfev, lm.max_steps = 50, None
        status = lm.run_device(x0)
        res = lm.result(status)
        x, grad = prob.lm_result(lm.cur)
        np.testing.assert_array_equal(x, prob.get_params(lm.cur))
        red = prob.get_reduced()
        gcam = np.zeros(12 * C)
        gcam[prob.cam_index] = red["gc"]
        np.testing.assert_array_equal(grad, np.concatenate([gcam, prob.frame_gradient().ravel()]))
        np.testing.assert_array_equal(res.lm["grad"], grad)
        x2, g2 = prob.lm_result(lm.cur, lazy_grad=True)      # the gradient left on the device as an object of its own
        np.testing.assert_array_equal(x2, x)
        prob.close()                                          # ... which outlives the handle
        np.testing.assert_array_equal(g2.download(), grad)


def test_nonfinite_start_is_scipys_error(mc):
    p = mc.synth.make_problem(2, 30, seed=1)
    bad = p["poses"].copy()
    bad[3, 5] = np.inf
    with pytest.raises(ValueError, match="Residuals are not finite in the initial point"):
        captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], bad, n_frames=None, outlier_threshold=1e300, verbose=0)


def test_skew_in_the_input_intrinsics_is_refused(mc):
    p = mc.synth.make_problem(2, 30, seed=1)
    intr = [(K.copy(), d) for K, d in p["intrinsics"]]
    intr[1][0][0, 1] = 0.3
    with pytest.raises(ValueError, match="skew"):
        mc.bundle_adjust(p["uvs"], p["extrinsics"], intr, p["obj"], p["poses"], n_frames=None, verbose=0)


# ------------------------------------------------------------------ bounds= (the reference forwards it to scipy's bounded TRF)
@pytest.mark.parametrize("tag", ["config1", "missing3"])
def test_bounds_reach_the_reference_bounded_optimum(mc, golden, tag):
    """Golden = the unmodified reference's bundle_adjust(..., bounds=(lo, hi)) polished to the constrained optimum and certified by
    finite differences (tests/golden/make_golden_bounds.py).  config1 (BASELINE configs[0]): ten bounds active there, five on camera
    parameters (k1, k2, fx) and five on board-pose coordinates; missing3 (three cameras, a quarter of the detections missing): six, one
    on a focal length and five on poses.  The GPU's active-set loop must end at the same point: cost to 1e-9, the same active set, every
    bounded coordinate ON its bound, parameters to 1e-6 relative; every iterate feasible; unbounded calls untouched (bit-identical)."""
    from conftest import problem_from_npz

    z = golden(f"tight_bounds_{tag}.npz")
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    lo, hi = z["lo"], z["hi"]
    (e, i, p_, use, res), out = captured(mc.bundle_adjust, uvs, ext, intr, obj, poses, n_frames=None, bounds=(lo, hi), ftol=1e-15, xtol=1e-15, gtol=1e-9, verbose=1, max_nfev=400)
    np.testing.assert_array_equal(use, z["use"])
    assert res.status in (1, 2, 3, 4), out
    assert np.all(res.x >= lo) and np.all(res.x <= hi)
    assert abs(res.cost - float(z["cost"])) <= 1e-9 * res.cost, (res.cost, float(z["cost"]))
    # The constrained minimiser is a MANIFOLD: the active pose bounds take five of the six gauge freedoms, along the sixth the cost is flat and
    # further pose bounds may or may not be touched.  The active set on the gauge-invariant parameters (the intrinsics) must be the golden's;
    # on config1 the whole set is (both runs stop at the same point of the manifold); in general every bound the golden has active is
    # active here, and what is active here beyond it is checked by the KKT conditions below.
    C_ = uvs.shape[0]
    intr_idx = np.array([12 * c + k for c in range(C_) for k in range(6)])
    np.testing.assert_array_equal(res.active_mask[intr_idx], z["active_mask"][intr_idx])
    if tag == "config1":
        np.testing.assert_array_equal(res.active_mask, z["active_mask"])
    act = res.active_mask != 0
    assert act.sum() >= 2
    np.testing.assert_array_equal(res.x[res.active_mask == -1], lo[res.active_mask == -1])
    np.testing.assert_array_equal(res.x[res.active_mask == 1], hi[res.active_mask == 1])
    # parameters to 1e-6 relative: the intrinsics directly; extrinsics and poses through their gauge-invariant combinations (camera-from-camera
    # and camera-from-board transforms) -- five active pose bounds take five of the six gauge freedoms, one is left
    xg, C = z["x"], uvs.shape[0]
    cam, cam_g = res.x[:12 * C].reshape(C, 12), xg[:12 * C].reshape(C, 12)
    assert (np.abs(cam[:, :6] - cam_g[:, :6]) / np.abs(cam_g[:, :6])).max() < 1e-6
    ext_a, _, poses_a = orc.deserialize_params(res.x, C)
    ext_g, _, poses_g = orc.deserialize_params(xg, C)
    (cc, cb), (cc_g, cb_g) = orc.invariants(ext_a, poses_a), orc.invariants(ext_g, poses_g)
    for a, b in ((cc, cc_g), (cb, cb_g)):
        assert np.abs(a - b)[..., :3, :3].max() < 1e-6
        assert (np.abs(a - b)[..., :3, 3] / np.abs(b[..., :3, 3]).max()).max() < 1e-6
    # the oracle's view of the returned point: its robust cost is the reported one, its gradient vanishes off the active set and points outward on it
    f = orc.residuals(res.x, uvs[:, use], obj)
    assert abs(orc.robust_cost(f) - res.cost) <= 1e-10 * res.cost
    js, fs = orc.robust_scales(f)
    g = orc.jacobian_csr(res.x, uvs[:, use], obj).T @ (js * fs)
    assert np.abs(g[~act]).max() < 1e-6 * np.abs(g[act]).max()
    gtol_ = 1e-6 * np.abs(g[act]).max()   # (a bound touched along the flat gauge direction carries a zero multiplier)
    assert np.all(g[res.active_mask == -1] > -gtol_) and np.all(g[res.active_mask == 1] < gtol_)
    np.testing.assert_allclose(res.grad, g, rtol=0, atol=1e-6 * np.abs(g).max())
    # infinite bounds = scipy's default = the unconstrained solver, bit for bit; a Bounds object and scalars are accepted
    from scipy.optimize import Bounds

    (_, _, _, _, r0), _ = captured(mc.bundle_adjust, uvs, ext, intr, obj, poses, n_frames=None, verbose=0)
    (_, _, _, _, r1), _ = captured(mc.bundle_adjust, uvs, ext, intr, obj, poses, n_frames=None, verbose=0, bounds=(-np.inf, np.inf))
    (_, _, _, _, r2), _ = captured(mc.bundle_adjust, uvs, ext, intr, obj, poses, n_frames=None, verbose=0, bounds=Bounds(np.full(xg.size, -np.inf), np.full(xg.size, np.inf)))
    np.testing.assert_array_equal(r0.x, r1.x)
    np.testing.assert_array_equal(r0.x, r2.x)
    assert not r0.active_mask.any()


def test_bounds_reach_the_reference_bounded_optimum_at_baseline_size(mc, golden):
    """Round 6: the same pin at a BASELINE size, 6 cameras x 1 000 frames x 54 points (tests/golden/make_golden_bounds_large.py): the unmodified
    reference's bounded run (scipy's trf_bounds through its **opt_kwargs -- still at optimality 1.3e6 after 80 evaluations and 16 minutes)
    polished by a sparse active-set Gauss-Newton iteration on the reference's residuals to a KKT residual of 4e-6 and certified by the finite-
    difference gradient of the reference's cost; 23 bounds active there (k1 / k2 / fx of three cameras, the z-translation of 18 board poses)."""
    z = golden("tight_bounds_6x1000.npz")
    C, F, N = (int(v) for v in z["shape"])
    p = mc.synth.make_problem(C, F, seed=0, perturb_seed=1)
    assert abs(float(z["uvs_checksum"]) - np.nansum(p["uvs"])) <= 1e-9 * abs(float(z["uvs_checksum"]))
    lo, hi, xg = z["lo"], z["hi"], z["x"]
    (e, i, p_, use, res), out = captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, bounds=(lo, hi), ftol=1e-15, xtol=1e-15, gtol=1e-9, verbose=1,
                                         max_nfev=400, return_jac=False)
    np.testing.assert_array_equal(use, z["use"])
    assert res.status in (1, 2, 3, 4), out
    assert np.all(res.x >= lo) and np.all(res.x <= hi)
    assert abs(res.cost - float(z["cost"])) <= 1e-9 * res.cost, (res.cost, float(z["cost"]))
    intr_idx = np.array([12 * c + k for c in range(C) for k in range(6)])
    np.testing.assert_array_equal(res.active_mask[intr_idx], z["active_mask"][intr_idx])
    gold_act = z["active_mask"] != 0
    np.testing.assert_array_equal(res.active_mask[gold_act], z["active_mask"][gold_act])   # every bound the golden has active is active here
    np.testing.assert_array_equal(res.x[res.active_mask == -1], lo[res.active_mask == -1])
    np.testing.assert_array_equal(res.x[res.active_mask == 1], hi[res.active_mask == 1])
    cam, cam_g = res.x[:12 * C].reshape(C, 12), xg[:12 * C].reshape(C, 12)
    assert (np.abs(cam[:, :6] - cam_g[:, :6]) / np.abs(cam_g[:, :6])).max() < 1e-6
    ext_a, _, poses_a = orc.deserialize_params(res.x, C)
    ext_g, _, poses_g = orc.deserialize_params(xg, C)
    (cc, cb), (cc_g, cb_g) = orc.invariants(ext_a, poses_a), orc.invariants(ext_g, poses_g)
    for a, b in ((cc, cc_g), (cb, cb_g)):
        assert np.abs(a - b)[..., :3, :3].max() < 1e-6
        assert (np.abs(a - b)[..., :3, 3] / np.abs(b[..., :3, 3]).max()).max() < 1e-6
    # KKT by the oracle at the returned point
    act = res.active_mask != 0
    f = orc.residuals(res.x, p["uvs"][:, use], p["obj"])
    js, fs = orc.robust_scales(f)
    g = orc.jacobian_csr(res.x, p["uvs"][:, use], p["obj"]).T @ (js * fs)
    assert np.abs(g[~act]).max() < 1e-6 * np.abs(g[act]).max()
    gtol_ = 1e-6 * np.abs(g[act]).max()
    assert np.all(g[res.active_mask == -1] > -gtol_) and np.all(g[res.active_mask == 1] < gtol_)


def test_bounds_with_fixed_intrinsics_and_validation(mc):
    """Bounds together with fix_intrinsics (flags on the 12-wide camera block): a tight box around the START values of every camera's
    extrinsics (single bounded coordinates would be evaded through the gauge freedom) -- the cameras cannot reach their unconstrained
    places, several bounds end active, the poses adapt.  Judged by the oracle: feasible, KKT point, intrinsics untouched."""
    p = mc.synth.make_problem(3, 40, seed=21)
    x0 = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    (_, _, _, use, ru), _ = captured(mc.bundle_adjust, *args, n_frames=None, verbose=0, fix_intrinsics=True, ftol=1e-14, xtol=1e-14, gtol=1e-10)
    lo, hi = np.full(x0.size, -np.inf), np.full(x0.size, np.inf)
    ext_idx = np.array([12 * c + k for c in range(3) for k in range(6, 12)])
    delta = np.tile(np.r_[np.full(3, 1e-4), np.full(3, 0.05)], 3)
    lo[ext_idx], hi[ext_idx] = x0[ext_idx] - delta, x0[ext_idx] + delta
    (_, intr_b, _, _, rb), _ = captured(mc.bundle_adjust, *args, n_frames=None, verbose=0, fix_intrinsics=True, bounds=(lo, hi), ftol=1e-14, xtol=1e-14, gtol=1e-10, max_nfev=400)
    assert rb.status > 0 and rb.cost > ru.cost * (1 + 1e-9) and np.all(rb.x >= lo) and np.all(rb.x <= hi)
    cams0 = x0[:36].reshape(3, 12)
    np.testing.assert_array_equal(rb.x[:36].reshape(3, 12)[:, :6], cams0[:, :6])   # intrinsics untouched
    act = rb.active_mask != 0
    assert set(np.nonzero(act)[0]) <= set(ext_idx) and act.sum() >= 2
    f = orc.residuals(rb.x, p["uvs"][:, use], p["obj"])
    assert abs(orc.robust_cost(f) - rb.cost) <= 1e-10 * rb.cost
    js, fs = orc.robust_scales(f)
    g = orc.jacobian_csr(rb.x, p["uvs"][:, use], p["obj"]).T @ (js * fs)
    free = np.ones(x0.size, bool)
    free[:36] = np.tile(np.r_[np.zeros(6, bool), np.ones(6, bool)], 3)
    free[act] = False
    assert np.abs(g[free]).max() < 1e-6 * np.abs(g[act]).max()
    assert np.all(g[rb.active_mask == -1] > 0) and np.all(g[rb.active_mask == 1] < 0)
    # scipy's checks, scipy's messages
    for bad, msg in (((lo, hi, hi), "must contain 2 elements"), ((lo[:5], hi), "Inconsistent shapes"), ((hi, lo), "strictly less"), ((x0 + 1.0, x0 + 2.0), "outside of provided bounds")):
        with pytest.raises(ValueError, match=msg):
            captured(mc.bundle_adjust, *args, n_frames=None, verbose=0, bounds=bad)


# ------------------------------------------------------------------ more shapes and corner cases of the one-crossing paths
@pytest.mark.parametrize("shape", [(24, 3000, 10, 20, dict(missing=0.3, outlier_frames=20)), (6, 100000, 6, 9, dict(missing=0.05, outlier_frames=50)), (2, 1, 6, 9, {}),
                                   (3, 65, 1, 1, dict(missing=0.4)), (5, 127, 2, 3, dict(missing=0.9))])
def test_fused_prefilter_shapes(mc, shape):
    """The selection kernels where their work items, slots and candidate lists are cut differently: 4 800 rows of errors (24 cameras x 200
    points), 1 563 frame blocks (100 000 frames: ~26 k candidates, past the final pass's LDS list), one frame, a one-point board, nearly
    everything missing -- against the per-call pre-filter (same frames, same line, same RNG state)."""
    C, F, rows, cols, kw = shape
    p = mc.synth.make_problem(C, F, rows=rows, cols=cols, seed=7, **kw)
    for n_frames, thr in ((None, None), (max(1, F // 3), None), (None, 2.0)):
        a = _select(mc, p, n_frames, thr, 3)
        b = _select(mc, p, n_frames, thr, 3, MCBA_PREFILTER_FUSED=0)
        np.testing.assert_array_equal(a[0], b[0])
        assert a[1] == b[1] and a[2] == b[2], (a[1], b[1])


def test_prefilter_without_a_single_valid_error(mc):
    """Every detection missing: np.nanmedian of nothing is NaN, nothing is complete in two cameras, no frame is used (the reference
    then hands an empty problem to scipy: status 1 after one evaluation)."""
    p = mc.synth.make_problem(3, 20, seed=1)
    uvs = np.full_like(p["uvs"], np.nan)
    a = _select(mc, dict(p, uvs=uvs), None, None, 0)
    b = _select(mc, dict(p, uvs=uvs), None, None, 0, MCBA_PREFILTER_FUSED=0)
    assert a[0].size == 0 and b[0].size == 0 and a[1] == b[1] and "nan" in a[1]
    (e, i, ps, use, res), _ = captured(mc.bundle_adjust, uvs, p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0)
    assert use.size == 0 and res.status == 1 and res.nfev == 1 and ps.shape == (0, 6)


def test_lm_run_with_iteration_limits_and_bounded_with_x_scale(mc):
    p = mc.synth.make_problem(4, 90, seed=12, missing=0.1)
    x0 = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    out = {}
    for mode in ("0", "1"):
        with env(MCBA_HOST_LOOP=mode):
            for k in (0, 1, 3):
                prob = mc.ops.Problem(p["uvs"], p["obj"])
                res = mc.solver.lm_solve(prob, x0, ftol=0.0, xtol=0.0, gtol=0.0, max_iterations=k)
                out[mode, k] = (res.x.copy(), res.cost, res.nfev, res.status, res.lm["steps"], res.lm["history"])
                prob.close()
    for k in (0, 1, 3):
        a, b = out["0", k], out["1", k]
        np.testing.assert_array_equal(a[0], b[0])
        assert a[1:] == b[1:], (k, a[1:5], b[1:5])
        assert a[3] == 0 and a[4] == k
    # bounds together with a numeric x_scale (fixed damping matrix): still a KKT point of the same problem
    (_, _, _, use, ru), _ = captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0, ftol=1e-14, xtol=1e-14, gtol=1e-10)
    xs0 = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use])
    lo, hi = np.full(xs0.size, -np.inf), np.full(xs0.size, np.inf)
    for i in (4, 5, 12 + 4, 24 + 5, 36 + 0, 36 + 1):
        b = xs0[i] + 0.5 * (ru.x[i] - xs0[i])
        if ru.x[i] > xs0[i]:
            hi[i] = b
        else:
            lo[i] = b
    scale = np.where(np.arange(xs0.size) % 6 < 3, 1.0, 50.0)
    scale[:48] = np.tile([1000.0, 1000.0, 500.0, 500.0, 0.1, 0.05, 1.0, 1.0, 1.0, 100.0, 100.0, 100.0], 4)
    (_, _, _, _, rb), _ = captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0, bounds=(lo, hi), x_scale=scale,
                                   ftol=1e-15, xtol=1e-15, gtol=1e-9, max_nfev=600)
    (_, _, _, _, rc), _ = captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0, bounds=(lo, hi),
                                   ftol=1e-15, xtol=1e-15, gtol=1e-9, max_nfev=600)
    assert rb.status > 0 and rc.status > 0 and rb.active_mask.any()
    np.testing.assert_array_equal(rb.active_mask, rc.active_mask)
    assert abs(rb.cost - rc.cost) <= 1e-9 * rc.cost and rb.cost > ru.cost
    assert np.all(rb.x >= lo) and np.all(rb.x <= hi)


def test_three_crossing_stub_over_raw_ctypes(mc):
    """INTEGRATION.md section 2's stub, executed literally through ctypes.CDLL (no ops.py): mcba_prefilter -> mcba_create_subset ->
    mcba_lm_run -> mcba_lm_history -> mcba_lm_result reproduce api.bundle_adjust's frames, solution and gradient."""
    import ctypes

    p = mc.synth.make_problem(3, 120, seed=31, missing=0.15, outlier_frames=4)
    uvs, obj = np.ascontiguousarray(p["uvs"]), np.ascontiguousarray(p["obj"])
    C, F, N = uvs.shape[:3]
    lib = ctypes.CDLL(mc.ops.LIB_PATH)
    dp = ctypes.POINTER(ctypes.c_double)
    P = lambda a: a.ctypes.data_as(dp)
    lib.mcba_last_error.restype = ctypes.c_char_p
    lib.mcba_prefilter.argtypes = [ctypes.c_void_p, dp, dp, dp, ctypes.c_double, ctypes.POINTER(ctypes.c_ubyte), dp]
    lib.mcba_create_subset.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.mcba_lm_run.argtypes = [ctypes.c_void_p, dp, dp, ctypes.c_char_p, dp]
    lib.mcba_lm_history.argtypes = [ctypes.c_void_p, dp, ctypes.c_size_t]
    lib.mcba_lm_result.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp, ctypes.POINTER(ctypes.c_void_p)]
    lib.mcba_destroy.argtypes = [ctypes.c_void_p]

    def check(rc):
        assert rc == 0, lib.mcba_last_error().decode()

    h = ctypes.c_void_p()
    check(lib.mcba_create(ctypes.byref(h), C, F, N, 0))
    x_all = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    status, info = np.empty(F, np.uint8), np.empty(8)
    check(lib.mcba_prefilter(h, P(uvs), P(obj), P(x_all), float("nan"), status.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), P(info)))
    use0 = np.flatnonzero(status & 1)
    use_frames = use0[(status[use0] & 2) == 0]
    np.random.seed(4)
    n_frames = 60
    if not (n_frames is None or n_frames > len(use_frames)):
        use_frames = np.random.choice(use_frames, n_frames, replace=False)
    sub = ctypes.c_void_p()
    idx = np.ascontiguousarray(use_frames, dtype=np.int32)
    check(lib.mcba_create_subset(ctypes.byref(sub), h, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(idx)))
    n = 12 * C + 6 * len(idx)
    opt = np.array([1e-4, 1e-8, 1e-8, 1e-3, 1e-9, 1e12, 0.1, 1.0, 1e-2, 100 * n, -1, 2, 0], dtype=np.float64)
    summary = np.zeros(4)
    check(lib.mcba_lm_run(sub, None, P(opt), None, P(summary)))
    rows = np.empty((int(summary[1]), 32))
    check(lib.mcba_lm_history(sub, P(rows), len(rows)))
    out = np.empty((2, n))
    check(lib.mcba_lm_result(sub, int(rows[-1, 3]), P(out), P(out[1]), None))
    lib.mcba_destroy(sub)
    lib.mcba_destroy(h)
    np.random.seed(4)
    (e, i, ps, use, res), line = captured(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=n_frames, verbose=0)
    np.testing.assert_array_equal(use, use_frames)
    assert f"threshold of {info[0]}" in line
    assert int(summary[0]) == res.status and 1 + int(rows[-1, 17]) == res.nfev
    np.testing.assert_array_equal(out[0], res.x)
    np.testing.assert_array_equal(out[1], res.grad)


# ---------------------------------------------------------------------------------------------------------------------------------
# least_squares' CALLABLE `loss` (the reference forwards `loss` untouched: bundle_adjustment.py:301-313; scipy least_squares.py:160-227)
@pytest.mark.parametrize("shape", [(3, 70), (6, 1000), (14, 130)])
def test_callable_loss_builds_the_builtin_normal_equations(mc, shape):
    """soft_l1 written as a callable must give the normal equations, cost and gradient of the built-in name (the table of the caller's
    rho values against the kernel's own rho: two routes to the same numbers), free and with the intrinsics held fixed; 3 x 70, 6 x 1 000 (16 frame
    blocks per camera), 14 cameras (the 16-tile k_syrk)."""
    from losses import soft_l1_as_callable

    p = mc.synth.make_problem(shape[0], shape[1], seed=5, perturb_seed=2, missing=0.2, scalar_nans=4)
    x0 = mc.api.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    for width in (12, 6):
        got = []
        for loss in ("soft_l1", soft_l1_as_callable):
            prob = mc.ops.Problem(p["uvs"], p["obj"], loss=loss, f_scale=0.8)
            if width == 6:
                assert prob.set_camera_block(6)
            prob.set_params(0, x0)
            prob.linearize(0)
            red = {k: v.copy() for k, v in prob.reduce_fetch(0.0).items()}
            got.append((red, prob.frame_gradient().copy(), prob.cost(0)[0]))
            if callable(loss):   # the device-resident loops cannot call back: they refuse
                with pytest.raises(mc.ops.McbaError, match="tabulated loss"):
                    prob.lm_run(None, 1e-8, 1e-8, 1e-8, 1e-3, 1e-12, 1e12, 0.1, 1.0, 0.01, 10, 10, 2)
            prob.close()
        (ra, ga, ca), (rb, gb, cb) = got
        assert abs(ca - cb) <= 1e-13 * ca and abs(ra["scal"][0] - rb["scal"][0]) <= 1e-13 * ca
        for k in ("S0", "rhs", "diagU", "gc"):
            np.testing.assert_allclose(rb[k], ra[k], rtol=0, atol=1e-11 * np.abs(ra[k]).max(), err_msg=f"{k} width {width}")
        np.testing.assert_allclose(gb, ga, rtol=0, atol=1e-11 * np.abs(ga).max())


def test_callable_loss_reaches_the_reference_optimum(mc, golden):
    """Golden = the reference's residual function minimised with a callable loss that is none of scipy's five names (generalised Charbonnier,
    exponent 1/4, f_scale 0.7: tests/losses.py) to a tight, FD-certified optimum from two starts (tests/golden/make_golden.py --callable), + what
    the reference's own bundle_adjust(..., loss=<function>) returns with its default tolerances.  bundle_adjust here with the same function:
    cost to 1e-9, parameters to 1e-6; fun / jac / grad are scipy's (unscaled residuals, robust-rescaled rows, J^T f of the rescaled pair)."""
    from conftest import problem_from_npz
    from losses import charbonnier_quarter

    z = golden("tight_config1_callable.npz")
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    (e, i, p_, use, res), out = captured(mc.bundle_adjust, uvs, ext, intr, obj, poses, n_frames=None, loss=charbonnier_quarter, f_scale=0.7, ftol=1e-14, xtol=1e-14, gtol=1e-9, verbose=2, max_nfev=300)
    np.testing.assert_array_equal(use, z["s0_use"])
    assert res.status in (1, 2, 3, 4), out
    assert "Iteration" in out and res.lm["reduced_solver"] == "host"
    assert abs(res.cost - float(z["s0_cost"])) <= 1e-9 * res.cost, (res.cost, float(z["s0_cost"]))
    xg, C = z["s0_x"], uvs.shape[0]
    cam, cam_g = res.x[:12 * C].reshape(C, 12), xg[:12 * C].reshape(C, 12)
    assert (np.abs(cam[:, :6] - cam_g[:, :6]) / np.abs(cam_g[:, :6])).max() < 1e-6
    ext_a, _, poses_a = orc.deserialize_params(res.x, C)
    ext_g, _, poses_g = orc.deserialize_params(xg, C)
    (cc, cb), (cc_g, cb_g) = orc.invariants(ext_a, poses_a), orc.invariants(ext_g, poses_g)
    for a, b in ((cc, cc_g), (cb, cb_g)):
        assert np.abs(a - b)[..., :3, :3].max() < 1e-6
        assert (np.abs(a - b)[..., :3, 3] / np.abs(b[..., :3, 3]).max()).max() < 1e-6
    # the oracle's view of the returned point (scipy's own arithmetic with the same function)
    f = orc.residuals(res.x, uvs[:, use], obj)
    assert abs(orc.robust_cost(f, charbonnier_quarter, 0.7) - res.cost) <= 1e-10 * res.cost
    js, fs = orc.robust_scales(f, charbonnier_quarter, 0.7)
    J = orc.jacobian_csr(res.x, uvs[:, use], obj)
    g = J.T @ (js * fs)
    np.testing.assert_allclose(res.fun, f, rtol=0, atol=1e-9)
    np.testing.assert_allclose(res.grad, g, rtol=0, atol=1e-11 * (abs(J).T @ np.abs(js * fs)).max())   # (at the optimum the entries are what cancellation leaves of their terms)
    Js = J.multiply(js[:, None]).tocsr()
    assert res.jac.shape == Js.shape and np.array_equal(res.jac.indptr, Js.indptr)
    assert abs(res.jac - Js).max() <= 1e-9 * abs(Js).max()
    # ... and with the reference's default tolerances: at least as low as the reference's own run with this function (which stops early: ftol 1e-4)
    (_, _, _, _, r1), _ = captured(mc.bundle_adjust, uvs, ext, intr, obj, poses, n_frames=None, loss=charbonnier_quarter, f_scale=0.7, verbose=0)
    assert r1.cost <= float(z["ref_default_cost"]) * (1 + 1e-9) and r1.cost >= float(z["s0_cost"]) * (1 - 1e-9)


def test_callable_loss_contract_and_combinations(mc):
    """scipy's message for a function that returns the wrong shape; a callable with fix_intrinsics and with (inactive and active) bounds."""
    from losses import charbonnier_quarter, soft_l1_as_callable

    p = mc.synth.make_problem(2, 40, seed=3, perturb_seed=1)
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    with pytest.raises(ValueError, match="The return value of `loss` callable has wrong shape."):
        captured(mc.bundle_adjust, *args, n_frames=None, loss=lambda z: np.ones((2, z.size)), verbose=0)
    tol = dict(ftol=1e-13, xtol=1e-13, gtol=1e-9, verbose=0, n_frames=None, max_nfev=300)
    # the callable form of soft_l1 ends where the name does (different drivers: host-driven table loop / device-resident loop)
    (_, ia, _, _, ra), _ = captured(mc.bundle_adjust, *args, loss="soft_l1", **tol)
    (_, ib, _, _, rb), _ = captured(mc.bundle_adjust, *args, loss=soft_l1_as_callable, **tol)
    assert abs(ra.cost - rb.cost) <= 1e-10 * ra.cost
    for (Ka, da), (Kb, db) in zip(ia, ib):
        np.testing.assert_allclose(Kb, Ka, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(db, da, rtol=1e-5, atol=1e-9)
    # fix_intrinsics: the 6-wide camera block with the table
    (_, ic, _, _, rc), _ = captured(mc.bundle_adjust, *args, loss=charbonnier_quarter, fix_intrinsics=True, **tol)
    (_, id_, _, _, rd), _ = captured(mc.bundle_adjust, *args, loss=charbonnier_quarter, fix_intrinsics=True, **dict(tol, ftol=1e-15, xtol=1e-15))
    for (Kc, dc), (K0, d0) in zip(ic, p["intrinsics"]):
        np.testing.assert_array_equal(Kc, K0)
        np.testing.assert_array_equal(dc[:2], np.asarray(d0)[:2])
    assert rc.x.size == 12 * 2 + 6 * 40   # (every frame is used: no detection is missing and none is an outlier)
    f = orc.residuals(rc.x, p["uvs"], p["obj"])
    assert abs(orc.robust_cost(f, charbonnier_quarter) - rc.cost) <= 1e-10 * rc.cost and abs(rc.cost - rd.cost) <= 1e-9 * rc.cost
    # bounds: a box nothing touches changes nothing but the driver; a box on k1 that the optimum violates ends ON it
    (_, _, _, _, r_free), _ = captured(mc.bundle_adjust, *args, loss=charbonnier_quarter, **tol)
    n = r_free.x.size
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    lo[0], hi[0] = r_free.x[0] - 1e4, r_free.x[0] + 1e4
    (_, _, _, _, r_wide), _ = captured(mc.bundle_adjust, *args, loss=charbonnier_quarter, bounds=(lo, hi), **tol)
    assert abs(r_wide.cost - r_free.cost) <= 1e-9 * r_free.cost and not r_wide.active_mask.any()
    x0 = mc.api.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    k = 4   # k1 of camera 0: a bound half-way between the start and the free optimum
    b = x0[k] + 0.5 * (r_free.x[k] - x0[k])
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    if r_free.x[k] > x0[k]:
        hi[k] = b
    else:
        lo[k] = b
    (_, _, _, _, r_box), _ = captured(mc.bundle_adjust, *args, loss=charbonnier_quarter, bounds=(lo, hi), **tol)
    assert r_box.x[k] == b and r_box.active_mask[k] != 0 and r_box.cost > r_free.cost
    f = orc.residuals(r_box.x, p["uvs"], p["obj"])
    js, fs = orc.robust_scales(f, charbonnier_quarter)
    g = orc.jacobian_csr(r_box.x, p["uvs"], p["obj"]).T @ (js * fs)
    free = r_box.active_mask == 0
    assert np.abs(g[free]).max() < 1e-5 * max(1.0, abs(g[k]))


def test_bounds_at_a_thousand_frames(mc):
    """The bounded loop where the frozen frame coordinates span many workgroups (6 x 1 000 x 54: 16 frame blocks per camera, the XS instances of
    k_syrk / k_backsub over all of them, k_clip over 6 072 coordinates): bounds on four intrinsics and on forty pose coordinates that the free optimum
    violates; judged by the oracle's KKT conditions at the returned point."""
    from scipy.optimize._lsq.common import find_active_constraints

    p = mc.synth.make_problem(6, 1000, seed=0, perturb_seed=1)
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    tol = dict(n_frames=None, ftol=1e-13, xtol=1e-13, gtol=1e-10, verbose=0, max_nfev=300)
    (_, _, _, use, free), _ = captured(mc.bundle_adjust, *args, **tol)
    assert use.size == 1000 and free.status > 0
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    xu, n, nc = free.x, x0.size, 72
    rng = np.random.default_rng(3)
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    pick = [4, 12 + 5, 24 + 0, 36 + 4] + list(nc + rng.choice(6000, 40, replace=False))
    for i in pick:
        b = x0[i] + (0.5 if i < nc else 0.1) * (xu[i] - x0[i])
        if xu[i] > x0[i]:
            hi[i] = b
        else:
            lo[i] = b
    (_, _, _, use2, res), _ = captured(mc.bundle_adjust, *args, bounds=(lo, hi), **tol)
    assert res.status > 0 and np.array_equal(use2, use)
    x = res.x
    assert np.all(x >= lo) and np.all(x <= hi) and res.cost >= free.cost * (1 - 1e-12)
    am = find_active_constraints(x, lo, hi, rtol=1e-13)
    np.testing.assert_array_equal(res.active_mask, am)
    assert (am[:nc] != 0).sum() >= 2 and (am[nc:] != 0).sum() >= 6
    f = orc.residuals(x, p["uvs"], p["obj"])
    assert abs(orc.robust_cost(f) - res.cost) <= 1e-10 * res.cost
    js, fs = orc.robust_scales(f)
    g = orc.jacobian_csr(x, p["uvs"], p["obj"]).T @ (js * fs)
    js0, fs0 = orc.robust_scales(orc.residuals(x0, p["uvs"], p["obj"]))
    scale = np.abs(orc.jacobian_csr(x0, p["uvs"], p["obj"]).T @ (js0 * fs0)).max()
    assert np.abs(g[am == 0]).max() <= 1e-6 * scale, (np.abs(g[am == 0]).max(), scale)
    assert np.all(g[am == -1] >= -1e-6 * scale) and np.all(g[am == 1] <= 1e-6 * scale)
    np.testing.assert_allclose(res.grad, g, rtol=0, atol=1e-6 * scale)


@pytest.mark.parametrize("shape", [(3, 70, dict(missing=0.2, outlier_frames=4)), (6, 300, dict(outlier_frames=7)), (2, 40, {})])
def test_prefilter_subset_equals_the_two_crossings(mc, shape):
    """mcba_prefilter_subset (the kept frames gathered inside the pre-filter's crossing when no random draw stands in between: n_frames None or
    larger than what is kept) against mcba_prefilter + mcba_create_subset (MCBA_PREFILTER_SUBSET=0): the same frames, printed line, iterates and
    solution to the bit; with n_frames <= the frames kept the reference's draw from the global RNG comes first and the old path runs (same result,
    same RNG state); nothing kept: no handle."""
    C, F, kw = shape
    p = mc.synth.make_problem(C, F, seed=21, perturb_seed=3, **kw)
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    for nf in (None, 10 * F, F // 2):
        outs = []
        for flag in ("1", "0"):
            with env(MCBA_PREFILTER_SUBSET=flag):
                np.random.seed(5)
                (e, i, ps, use, res), out = captured(mc.bundle_adjust, *args, n_frames=nf, verbose=2, return_jac=False)
                outs.append((use, res.x, res.cost, res.nfev, out, np.random.randint(1 << 30)))
        a, b = outs
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and a[5] == b[5]
    # ... with the options that take other drivers: a callable loss, bounds, the intrinsics held fixed
    from losses import charbonnier_quarter

    for extra in (dict(loss=charbonnier_quarter, f_scale=0.7), dict(fix_intrinsics=True), dict(bounds=(-1e9, 1e9), x_scale=2.0)):
        outs = []
        for flag in ("1", "0"):
            with env(MCBA_PREFILTER_SUBSET=flag):
                (e, i, ps, use, res), out = captured(mc.bundle_adjust, *args, n_frames=None, verbose=0, return_jac=False, max_nfev=30, **extra)
                outs.append((use, res.x, res.cost))
        np.testing.assert_array_equal(outs[0][0], outs[1][0])
        np.testing.assert_array_equal(outs[0][1], outs[1][1])
        assert outs[0][2] == outs[1][2]
    # the crossing itself: which case it reports, and the handle it makes
    x = mc.api.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = mc.ops.Problem(p["uvs"], p["obj"], upload=False)
    status, thr, info, sub = prob.prefilter_subset(x, None, None, "soft_l1", 1.0)
    kept = np.flatnonzero((status & 3) == 1)
    if kept.size == F:
        assert info[7] == 2 and sub is None
    else:
        assert info[7] == 3 and sub is not None and sub.F == kept.size
        np.testing.assert_array_equal(sub.get_params(0), np.concatenate([x[:12 * C], x[12 * C:].reshape(F, 6)[kept].ravel()]))
        np.testing.assert_array_equal(sub.download_observations(), p["uvs"][:, kept])
        sub.close()
    status2, thr2, info2, sub2 = prob.prefilter_subset(x, None, max(1, kept.size // 2), "soft_l1", 1.0)
    assert info2[7] == 1 and sub2 is None and np.array_equal(status2, status)
    prob.close()
    q = mc.synth.make_problem(2, 12, seed=3)
    q["uvs"][1] = np.nan   # no frame is complete in two cameras
    prob = mc.ops.Problem(q["uvs"], q["obj"], upload=False)
    st, _, info3, sub3 = prob.prefilter_subset(mc.api.serialize_params(q["extrinsics"], q["intrinsics"], q["poses"]), None, None, "soft_l1", 1.0)
    assert info3[7] == 0 and sub3 is None and not (st & 1).any()
    prob.close()
