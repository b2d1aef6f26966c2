"""GPU tests of the round-3 additions to the C ABI: buffer pool + lazily allocated solver buffers, the residual vector that stays on
the device until `result.fun` is read, least_squares' numeric x_scale, the sharded radix select of the pre-filter, and the
observable poll timeout of the fused back-substitution.  Run with `-m gpu` on an MI355X."""
import contextlib
import io
import os

import numpy as np
import pytest

from fake_problem import OracleProblem
from oracle import ba_oracle as orc

pytestmark = pytest.mark.gpu


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


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


# ------------------------------------------------------------------ result.fun: on the device until read; handles come from the pool
def test_lazy_fun_outlives_the_handle_and_matches_the_oracle(mc):
    p = mc.synth.make_problem(3, 70, seed=5, missing=0.2, scalar_nans=4)
    kw = dict(n_frames=None, ftol=1e-10, verbose=0, return_jac=False)
    e, it, ps, use, res = quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)
    assert isinstance(dict.get(res, "fun"), mc.api._Lazy)           # nothing downloaded yet; the Problem is closed by now
    again = [quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)[4] for _ in range(3)]   # pooled buffers are re-used
    f = orc.residuals(res.x, p["uvs"][:, use], p["obj"])
    np.testing.assert_allclose(res.fun, f, rtol=0, atol=1e-9)        # NaN scalars removed, (C,F,N,2) order
    assert not isinstance(dict.get(res, "fun"), mc.api._Lazy)
    for r in again:
        np.testing.assert_array_equal(r.x, res.x)                     # recycled (not re-zeroed) buffers change nothing
        np.testing.assert_array_equal(r.fun, res.fun)
    full = quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, ftol=1e-10, verbose=0)[4]
    assert isinstance(dict.get(full, "jac"), mc.api._Lazy)          # the default call (return_jac=True) does not pay for the Jacobian either ...
    np.testing.assert_array_equal(full.fun, res.fun)                 # ... and reading `fun` does not produce it
    assert isinstance(dict.get(full, "jac"), mc.api._Lazy)
    J = full.jac                                                     # produced now, from the handle the result kept alive
    assert J.shape == (res.fun.size, res.x.size) and J.nnz == 18 * res.fun.size
    z = res.fun ** 2
    rho1 = (1 + z) ** -0.5                                           # soft_l1: rho' ; scipy's row scaling js = sqrt(rho' + 2 rho'' f^2) = (1 + z)^-3/4
    js = (1 + z) ** -0.75
    np.testing.assert_allclose(J.T @ (rho1 * res.fun / js), full.grad, rtol=0, atol=1e-8 * max(1.0, np.abs(full.grad).max()) + 1e-7)
    assert full.jac is J                                             # an ordinary field from here on
    mc.ops.pool_trim()
    r2 = quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)[4]
    np.testing.assert_array_equal(r2.x, res.x)


def test_prefilter_handle_allocates_no_solver_buffers(mc):
    """a handle that only scores frames holds the observations, not 2 x F x C x 800 B of records."""
    import torch

    p = mc.synth.make_problem(6, 20000, seed=1)
    mc.ops.pool_trim()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    prob = mc.ops.Problem(p["uvs"], p["obj"])
    prob.set_params(0, mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"]))
    mean, full = prob.frame_errors(0)
    med = prob.error_median(None)[0]
    used = free0 - torch.cuda.mem_get_info()[0]
    obs = 2 * 6 * 20000 * 54 * 16
    assert used < obs + 0.6 * obs, (used, obs)      # two observation layouts + the per-point errors; records alone would add 2 x 96 MB
    assert np.isfinite(med) and mean.shape == (6, 20000)
    prob.linearize(0)                                # the first solver call allocates the rest
    assert free0 - torch.cuda.mem_get_info()[0] > used + 150e6
    prob.close()


# ------------------------------------------------------------------ sharded radix select == the single-handle median
@pytest.mark.parametrize("shards", [2, 3])
def test_error_histograms_of_frame_shards_give_the_exact_median(mc, shards):
    p = mc.synth.make_problem(4, 333, seed=11, missing=0.25, scalar_nans=9)
    x = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    mask = np.random.default_rng(3).uniform(size=333) < 0.7
    whole = mc.ops.Problem(p["uvs"], p["obj"])
    whole.set_params(0, x)
    whole.frame_errors(0)
    want = whole.error_median(mask)[0]
    whole.close()
    bounds = mc.api._split_bounds(333, shards)
    probs = []
    for r in range(shards):
        lo, hi = bounds[r], bounds[r + 1]
        q = mc.ops.Problem(p["uvs"][:, lo:hi], p["obj"])
        q.set_params(0, np.concatenate([x[:48], x[48 + 6 * lo:48 + 6 * hi]]))
        q.frame_errors(0)
        probs.append((q, mask[lo:hi]))
    first = {"v": True}

    def hist(prefix, pas):
        h = sum(q.error_histogram(m if first["v"] else None, prefix, pas).astype(np.int64) for q, m in probs)
        first["v"] = False
        return h

    got = mc.api._median_from_histograms(hist)
    for q, _ in probs:
        q.close()
    assert got == want                                # bit for bit: integer counts, whatever the sharding
    err = np.sqrt(((p["uvs"] - orc.predict_from_x(x, 4, p["obj"])) ** 2).sum(-1))[:, mask]
    assert abs(got - np.nanmedian(err)) <= 1e-10


# ------------------------------------------------------------------ numeric x_scale: same iterates as the oracle-driven LM
@pytest.mark.parametrize("reduced_solver", ["device", "host"])
@pytest.mark.parametrize("scalar", [False, True])
def test_numeric_x_scale_matches_the_oracle_driven_lm(mc, reduced_solver, scalar):
    """least_squares' x_scale as a fixed damping matrix D = 1 / x_scale^2 (bundle_adjustment.py:301-304 forwards it): the GPU
    loop against the same LM driver on the CPU oracle (tests/fake_problem.py) -- the same sequence of trial costs and dampings."""
    C, F = 3, 40
    p = mc.synth.make_problem(C, F, seed=21, missing=0.15)
    x0 = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    if scalar:
        xs = np.full(x0.size, 0.5)
    else:
        xs = np.concatenate([np.tile([100.0, 100.0, 50.0, 50.0, 0.05, 0.05, 0.01, 0.02, 0.01, 5.0, 4.0, 5.0], C), np.tile([0.01, 0.01, 0.02, 5.0, 5.0, 3.0], F)])
    kw = dict(ftol=0.0, xtol=1e-13, gtol=1e-9, max_nfev=60, x_scale=xs)
    ref = mc.solver.lm_solve(OracleProblem(p["uvs"], p["obj"]), x0, **kw)
    prob = mc.ops.Problem(p["uvs"], p["obj"])
    got = mc.solver.lm_solve(prob, x0, reduced_solver=reduced_solver, **kw)
    jac = mc.solver.lm_solve(prob, x0, reduced_solver=reduced_solver, ftol=0.0, xtol=1e-13, gtol=1e-9, max_nfev=60)   # back to 'jac' on the same handle
    prob.close()
    ha, hb, hj = np.array(ref.lm["history"]), np.array(got.lm["history"]), np.array(jac.lm["history"])
    # the steps that still change the cost by more than round-off (the loop converges in a handful of steps; with ftol = 0 it then runs on at noise level, where gain ratios -- and with them the dampings -- are not reproducible between two implementations)
    live = int(np.argmax(~((ha[:, 1] - ha[:, 2]) > 1e-9 * ha[:, 1]))) if (~((ha[:, 1] - ha[:, 2]) > 1e-9 * ha[:, 1])).any() else len(ha)
    n = min(len(ha), len(hb), 12, live)
    assert n >= 4
    np.testing.assert_allclose(hb[:n, 1:3], ha[:n, 1:3], rtol=1e-6)    # cost before / after every trial step (far from the optimum round-off differences grow: 1e-8 observed)
    np.testing.assert_allclose(hb[:n, 5], ha[:n, 5], rtol=1e-4)        # the damping schedule (a function of the gain ratios)
    assert not np.allclose(hj[:3, 2], hb[:3, 2], rtol=1e-6)            # ... and it is not the 'jac' path
    assert abs(got.cost - ref.cost) <= 1e-9 * ref.cost and abs(jac.cost - ref.cost) <= 1e-9 * ref.cost


def test_bundle_adjust_accepts_x_scale_like_scipy(mc):
    p = mc.synth.make_problem(2, 30, seed=4)
    kw = dict(n_frames=None, verbose=0, return_jac=False, ftol=1e-12, xtol=1e-12, outlier_threshold=1e9)
    a = quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)[4]
    b = quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], x_scale=2.0, **kw)[4]
    assert abs(a.cost - b.cost) <= 1e-9 * a.cost
    with pytest.raises(ValueError, match="positive numbers"):
        quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], x_scale=-1.0, **kw)
    with pytest.raises(ValueError, match="Inconsistent shapes"):
        quiet(mc.bundle_adjust, p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], x_scale=np.ones(7), **kw)


# ------------------------------------------------------------------ a poll that runs out is observable and harmless
@pytest.mark.parametrize("shape", [(6, 300, {}), (3, 130, dict(missing=0.2, outlier_frames=4))])
def test_fused_backsub_poll_timeout_is_observable_and_harmless(mc, shape):
    """MCBA_FUSE_MAX_POLLS=0: every back-substitution workgroup of k_solve_backsub that does not find the solve's word at its
    first look gives up.  The stale trial points are discarded ON THE DEVICE (the next tick only rebuilds the system), the host
    switches to the two-launch path, and the iterates are those of the unfused loop to the last bit."""
    C, F, extra = shape
    p = mc.synth.make_problem(C, F, seed=80 + C, **extra)
    p["poses"][::7, 3:] += 40.0                                   # rejected steps and growing damping on the way
    x0 = mc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    out = {}
    for tag, e in (("timeout", dict(MCBA_FUSE_MAX_POLLS=0)), ("unfused", dict(MCBA_FUSE_BACKSUB=0)), ("fused", {})):
        with env(**e):
            prob = mc.ops.Problem(p["uvs"], p["obj"])
            res = mc.solver.lm_solve(prob, x0, ftol=1e-12, xtol=1e-12, gtol=1e-9, max_nfev=120)
            out[tag] = (res, prob.fuse_status())
            prob.close()
    (a, sa), (b, sb), (c, sc) = out["timeout"], out["unfused"], out["fused"]
    assert sa[0] > 0 and sa[1] is False                           # it happened, it was seen, the handle left the fused path
    assert sb == (0.0, False) and sc == (0.0, True)
    assert a.lm["fuse_timeout_tick"] > 0 and c.lm["fuse_timeout_tick"] == 0
    assert a.lm["rebuilds"] >= 1 and b.lm["rebuilds"] == 0        # the discarded trial points cost rebuild-only ticks, nothing else
    assert a.status == b.status and a.nfev == b.nfev and a.lm["iterations"] == b.lm["iterations"]
    np.testing.assert_array_equal(a.x, b.x)
    assert a.cost == b.cost
    np.testing.assert_array_equal(np.array(a.lm["history"]), np.array(b.lm["history"]))
    np.testing.assert_array_equal(c.x, b.x)


# ------------------------------------------------------------------ round 4: the lazy fields fix their mask at call time; pending Jacobians do not pin HBM without bound
def test_lazy_fields_keep_the_mask_of_call_time_and_release_handles_under_pressure(mc):
    p = mc.synth.make_problem(3, 70, seed=5, missing=0.2, scalar_nans=4)
    uvs = p["uvs"].copy()
    args = lambda: (uvs, p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    kw = dict(n_frames=None, ftol=1e-10, verbose=0)
    ref = quiet(mc.bundle_adjust, *args(), **kw)[4]
    Jref, fref = ref.jac.copy(), ref.fun.copy()
    # (i) the caller edits its array between the call and the first read (NaN-ing detections for a second pass): fun / jac are those of the call
    res = quiet(mc.bundle_adjust, *args(), **kw)[4]
    uvs[0, :5] = np.nan
    np.testing.assert_array_equal(res.fun, fref)
    J = res.jac
    assert J.shape == Jref.shape and (J != Jref).nnz == 0
    uvs[...] = p["uvs"]
    # (ii) dict(result) / {**result} / OptimizeResult(result) see resolved fields, not placeholders
    res = quiet(mc.bundle_adjust, *args(), **kw)[4]
    d = dict(res)
    assert isinstance(d["fun"], np.ndarray) and d["jac"].shape == Jref.shape
    # (iii) a sweep that keeps its results: with a hold budget of 0 no pending result owns a handle, and its Jacobian is still produced on demand
    with env(MCBA_JAC_HOLD_MB="0"):
        held = [quiet(mc.bundle_adjust, *args(), **kw)[4] for _ in range(3)]
        assert not mc.api._JacobianSource.live
        assert all(isinstance(dict.get(r, "jac"), mc.api._Lazy) for r in held)
        J2 = held[1].jac
    assert (J2 != Jref).nnz == 0
    # (iv) ADVICE r4: a result whose handle was released keeps the VALUES the solve saw (copied back from the GPU at release time), not a
    # reference to the caller's array -- which is edited in place here before the field is read
    with env(MCBA_JAC_HOLD_MB="0"):
        r4 = quiet(mc.bundle_adjust, *args(), **kw)[4]
        uvs[1, 3:9] += 25.0
        J4 = r4.jac
    uvs[...] = p["uvs"]
    assert (J4 != Jref).nnz == 0
    # ... and with the default budget the handle is kept (no second upload) until the field is read or the result dropped
    keep = quiet(mc.bundle_adjust, *args(), **kw)[4]
    assert len(mc.api._JacobianSource.live) == 1
    src = mc.api._JacobianSource.live[0]
    obs_bytes = 8 * int(np.prod(src.shape4))   # (the frames the solve ran on: the pre-filter dropped some)
    assert src.bytes == src.prob.device_bytes() and 2 * obs_bytes <= src.bytes < 3.5 * obs_bytes   # what is parked is what is counted: both observation layouts + parameters, the solver buffers are gone (trim)
    del keep, src
    import gc

    gc.collect()
    assert not mc.api._JacobianSource.live


# ------------------------------------------------------------------ round 4: the presence bits of the observations come from the GPU's copy
@pytest.mark.parametrize("shape", [(2, 1, 1), (3, 7, 5), (2, 50, 54), (6, 333, 35), (5, 64, 16)])
def test_seen_bits_equal_numpy_packbits(mc, shape):
    C, F, N = shape   # scalar counts 4, 210, 10 800, 139 860, 10 240: below one wavefront, ragged last word / last byte, whole words
    rng = np.random.default_rng(C * 1000 + F)
    uvs = rng.normal(size=(C, F, N, 2))
    uvs[rng.random((C, F, N)) < 0.3] = np.nan          # whole detections ...
    uvs[rng.random((C, F, N, 2)) < 0.05] = np.nan      # ... and single scalars
    uvs[..., -1, :] = np.nan if F % 2 else uvs[..., -1, :]
    obj = rng.normal(size=(N, 3))
    prob = mc.ops.Problem(uvs, obj)
    try:
        np.testing.assert_array_equal(prob.seen_bits(), np.packbits(~np.isnan(uvs)))
        if F > 3:
            keep = np.array([F - 1, 0, 2])
            sub = prob.subset(keep)                       # (frames gathered on the device)
            try:
                np.testing.assert_array_equal(sub.seen_bits(), np.packbits(~np.isnan(uvs[:, keep])))
            finally:
                sub.close()
    finally:
        prob.close()
