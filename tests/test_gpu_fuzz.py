"""Randomised end-to-end checks of bundle_adjust() on the GPU against the ORACLE's objective (run with `-m gpu` on an MI355X).

scipy's TRF needs hundreds of evaluations to reach a tight optimum (the committed goldens were made that way, offline), so the seeded
random cases here are judged by properties the reference's minimiser has, each evaluated with the oracle (numpy restatement of
bundle_adjustment.py) at the point the GPU returns:
  * the frames used are the oracle's pre-filter selection;
  * `cost`, `fun` are the oracle's robust cost / residual vector at `x`; the cost did not go up;
  * `x` is a STATIONARY point of the oracle's objective: its gradient J^T rho'(f) (scipy's, common.py:720-731) is at the round-off floor
    of the initial gradient, and equals `result.grad`;
  * every case terminates (all five losses);
  * variants of the product that must not matter do not: the reduced camera system solved on the device or on the host, the 6-wide camera
    block or flags on the 12-wide one (fix_intrinsics), frame shards in one process (the north_star's partition) against one handle."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import ba_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import multicam_calibration_amd as m

    m.ops.load_library()
    return m


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


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


def draw(it):
    rng = np.random.default_rng(9000 + it)
    C = int(rng.choice([1, 2, 3, 4, 6, 9, 10, 14, 20]))
    F = int(rng.integers(3, 40)) if C > 6 else int(rng.integers(3, 200))
    rows, cols = int(rng.integers(2, 5)), int(rng.integers(2, 6))
    loss = str(rng.choice(["soft_l1", "linear", "huber", "cauchy", "arctan"]))
    opts = dict(loss=loss, f_scale=float(rng.choice([1.0, 0.5, 2.5])) if loss != "linear" else 1.0)
    mk = dict(n_cameras=C, n_frames=F, rows=rows, cols=cols, pitch=float(rng.choice([12.5, 40.0])), seed=700 + it, perturb_seed=800 + it,
              missing=float(rng.choice([0.0, 0.1, 0.3])), scalar_nans=int(rng.choice([0, 0, 7])), outlier_frames=int(rng.choice([0, 0, 2])) if F > 10 else 0)
    fixed = bool(rng.random() < 0.35) and C <= 26
    return mk, opts, fixed


EXTRA = int(os.environ.get("MCBA_FUZZ_EXTRA", "0"))   # a soak: that many further seeded cases per sweep (development aid; the committed suite runs the fixed ones)
CASES = list(range(64)) + list(range(1000, 1000 + EXTRA))


@pytest.mark.parametrize("it", CASES)
def test_returned_point_is_a_minimiser_of_the_oracle_objective(mc, it):
    mk, opts, fixed = draw(it)
    p = mc.synth.make_problem(**mk)
    C = mk["n_cameras"]
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    kw = dict(n_frames=None, ftol=1e-13, xtol=1e-13, gtol=1e-11, verbose=0, max_nfev=400, fix_intrinsics=fixed, **opts)
    tag = f"case {it}: {mk} {opts} fixed={fixed}"
    e, intr, poses, use, res = quiet(mc.bundle_adjust, *args, **kw)
    # every loss terminates (with Triggs' curvature alone -- rounds 1-3 -- eight of these 64 cases ran into max_nfev: redescending losses
    # started far from the optimum; the loop now starts on the IRLS weight, solver.py: CURV_SWITCH).  For the redescending losses
    # (cauchy, arctan) WHICH local minimum is reached may still depend on round-off, so stationarity to 1e-7 and the agreement of the
    # variants are asked of linear / soft_l1 / huber.
    tame = opts["loss"] in ("linear", "soft_l1", "huber")
    assert res.status > 0, tag
    # the selection: the oracle's pre-filter (bundle_adjustment.py:265-298)
    use_o = quiet(orc.prefilter_frames, *args, None, None)[0]
    np.testing.assert_array_equal(use, use_o, err_msg=tag)
    if use.size == 0:   # (one camera: no frame is complete in two -- the reference hands least_squares an empty problem)
        assert res.status == 1 and res.nfev == 1 and res.fun.size == 0 and res.cost == 0.0, tag
        np.testing.assert_array_equal(res.x, orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use]))
        return
    uvs = p["uvs"][:, use]
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use])
    x = res.x
    assert x.shape == x0.shape
    # cost / fun at x are the oracle's
    f = orc.residuals(x, uvs, p["obj"])
    cost = orc.robust_cost(f, opts["loss"], opts["f_scale"])
    assert abs(res.cost - cost) <= 1e-11 * cost + 1e-14, tag
    np.testing.assert_allclose(res.fun, f, rtol=0, atol=1e-9 * max(1.0, np.abs(f).max()), err_msg=tag)
    c0 = orc.robust_cost(orc.residuals(x0, uvs, p["obj"]), opts["loss"], opts["f_scale"])
    assert cost <= c0 * (1 + 1e-12), tag
    # stationarity in the oracle's gradient (free parameters), against the gradient at the start
    _, gc0, _, gf0, _, _ = orc.normal_equations(x0, uvs, p["obj"], opts["loss"], opts["f_scale"])
    _, gc, _, gf, _, _ = orc.normal_equations(x, uvs, p["obj"], opts["loss"], opts["f_scale"])
    if fixed:
        gc0, gc = gc0.copy(), gc.copy()
        gc0[:, :6], gc[:, :6] = 0.0, 0.0
        np.testing.assert_array_equal(x[:12 * C].reshape(C, 12)[:, :6], x0[:12 * C].reshape(C, 12)[:, :6], err_msg=tag)   # intrinsics untouched
    g0 = max(np.abs(gc0).max(), np.abs(gf0).max())
    g = np.concatenate([gc.ravel(), gf.ravel()])
    if tame:   # (a redescending loss can crawl into its ftol / xtol test well before the gradient vanishes)
        assert np.abs(g).max() <= 1e-7 * g0 + 1e-8, (tag, np.abs(g).max(), g0)
    assert np.abs(res.grad - g).max() <= 1e-6 * max(g0, np.abs(g).max()) + 1e-7, tag
    assert abs(res.optimality - np.abs(res.grad).max()) <= 1e-12 * g0 + 1e-300, tag
    # returned pieces are x, re-shaped the reference's way
    e2, i2, p2 = orc.deserialize_params(x, C)
    np.testing.assert_array_equal(e, e2)
    np.testing.assert_array_equal(poses, p2)
    for (K, d), (K2, d2) in zip(intr, i2):
        np.testing.assert_array_equal(K, K2)
        np.testing.assert_array_equal(d, d2)

    if not tame:
        return
    # ---- variants that must not matter (compared through what they predict: the rig's gauge is free)
    pred = orc.predict_from_x(x, C, p["obj"])
    others = {"host-solved camera system": dict(reduced_solver="host")}
    for name, extra in others.items():
        r = quiet(mc.bundle_adjust, *args, **kw, **extra)[4]
        assert r.status > 0 and abs(r.cost - res.cost) <= 1e-9 * res.cost + 1e-13, (tag, name)
        assert np.abs(orc.predict_from_x(r.x, C, p["obj"]) - pred).max() < 1e-4, (tag, name)
    if fixed:
        with env(MCBA_FIXED_COMPACT="0"):
            r = quiet(mc.bundle_adjust, *args, **kw)[4]
        assert r.status > 0 and abs(r.cost - res.cost) <= 1e-9 * res.cost + 1e-13, (tag, "flags on the 12-wide block")
        assert np.abs(orc.predict_from_x(r.x, C, p["obj"]) - pred).max() < 1e-4, (tag, "flags on the 12-wide block")


# The soak's unterminated cases (tests/tools/fuzz_scan.py 1000 2700, profiles/round6/fuzz_scan_r6a.txt: these 8 of 1 700): recordings of 3-8 frames
# -- the intrinsics and distortion of up to 20 cameras from three views of a 6-20-point board, barely determined.  Each run slides along a nearly
# flat, CURVED valley of the objective: accepted steps at a constant gain ratio of ~0.55 (Nielsen's rule then leaves the damping where it is), a
# relative gain of 1e-6 per step, the gradient not shrinking.  At the sweep's tolerances (1e-13) nothing fires within 400 evaluations -- and
# nothing fires in scipy's TRF either: the reference's own solver (same objective, analytic Jacobian, same budget) ends with status 0 at a
# HIGHER cost on every one of them (case 1660: 8.2243 against 8.0068 here; a geodesic-acceleration prototype needs 585 evaluations to get the
# gradient to 4e-3: profiles/round6/NOTES_round6.md section 5).  What is asserted is what holds: at the reference's default tolerance the runs
# stop (after ~110 evaluations at most; scipy: 21-103) with a cost at least as low as scipy's, and the tight runs descend monotonically below that and below scipy's tight run.
VALLEY_CASES = [1473, 1660, 1731, 1790, 1821, 1991, 2427, 2610]


@pytest.mark.parametrize("it", VALLEY_CASES)
def test_flat_valley_cases_are_no_worse_than_the_references_solver(mc, it):
    from scipy.optimize import least_squares

    mk, opts, fixed = draw(it)
    assert not fixed
    p = mc.synth.make_problem(**mk)
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    use = quiet(orc.prefilter_frames, *args, None, None)[0]
    uvs = p["uvs"][:, use]
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use])

    def scipy_run(**tol):   # the reference's call (bundle_adjustment.py:307-313) on the oracle's residuals, with an analytic Jacobian
        return least_squares(orc.residuals, x0, jac=lambda x, u, o: orc.jacobian_csr(x, u, o), x_scale="jac", method="trf", loss=opts["loss"], f_scale=opts["f_scale"], args=(uvs, p["obj"]), **tol)

    # the reference's default tolerance: both stop, ours at a cost no higher
    ref = scipy_run(ftol=1e-4, max_nfev=400)
    res = quiet(mc.bundle_adjust, *args, n_frames=None, verbose=0, return_jac=False, **opts)[4]
    assert ref.status > 0 and res.status > 0 and res.nfev <= 200, (it, ref.status, res.status, res.nfev)
    assert res.cost <= ref.cost * (1 + 1e-9), (it, res.cost, ref.cost)
    # the sweep's tolerances: neither terminates within 400 evaluations (status 0 is scipy's answer too); ours descends monotonically and further
    tight = quiet(mc.bundle_adjust, *args, n_frames=None, ftol=1e-13, xtol=1e-13, gtol=1e-11, verbose=0, max_nfev=400, return_jac=False, **opts)[4]
    ref_tight = scipy_run(ftol=1e-13, xtol=1e-13, gtol=1e-11, max_nfev=150)
    assert tight.cost <= res.cost and tight.cost <= ref_tight.cost * (1 + 1e-9), (it, tight.cost, res.cost, ref_tight.cost)
    hist = np.array([(h[1], h[2]) for h in tight.lm["history"]])
    accepted = hist[:, 1] <= hist[:, 0]
    assert accepted.sum() > 0.8 * len(hist)   # (a valley, not a fight with the damping: nearly every step is accepted)
    f = orc.residuals(tight.x, uvs, p["obj"])
    assert abs(orc.robust_cost(f, opts["loss"], opts["f_scale"]) - tight.cost) <= 1e-11 * tight.cost


@pytest.mark.parametrize("it", [2, 8, 14, 21, 23, 30, 37, 41, 48, 55, 62])
def test_frame_shards_in_one_process_reach_the_same_minimiser(mc, it):
    """The north_star's partition on random problems: the frames dealt out to 3 shards (solver.InProcessShards: one handle and one LM loop
    per shard, one rank-ordered sum of the reduced camera systems per iteration) against ONE handle -- same decisions, same optimum."""
    from multicam_calibration_amd import solver

    mk, opts, fixed = draw(it)
    mk = dict(mk, outlier_frames=0, n_frames=max(mk["n_frames"], 9))
    p = mc.synth.make_problem(**mk)
    C, F = mk["n_cameras"], mk["n_frames"]
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])

    import torch

    def one(frames, rank=None, shards=None):
        prob = mc.ops.Problem(np.ascontiguousarray(p["uvs"][:, frames]), p["obj"], loss=opts["loss"], f_scale=opts["f_scale"])
        try:
            if fixed:
                assert prob.set_camera_block(6)          # (before the collective's buffer is sized)
            comm = shards.comm(rank, prob, torch.device("cuda:0")) if shards is not None else None
            xs = np.concatenate([x0[:12 * C], x0[12 * C:].reshape(F, 6)[frames].ravel()])
            return solver.lm_solve(prob, xs, comm=comm, ftol=1e-13, xtol=1e-13, gtol=1e-11, max_nfev=300, verbose=0)
        finally:
            prob.close()

    whole = one(np.arange(F))
    world = 3
    parts = [np.arange(F)[r::world] for r in range(world)]
    shards = solver.InProcessShards(world)
    res = shards.run(lambda rank: one(parts[rank], rank, shards))
    key = lambda r: np.array([(h[0], h[1], h[2], h[5]) for h in r.lm["history"]])
    for r in res[1:]:
        np.testing.assert_array_equal(key(r), key(res[0]))                       # every shard takes the same decisions, to the bit
        np.testing.assert_array_equal(r.x[:12 * C], res[0].x[:12 * C])
    assert res[0].status > 0 and whole.status > 0
    assert abs(res[0].cost - whole.cost) <= 1e-9 * whole.cost + 1e-13
    xg = np.empty_like(x0)
    xg[:12 * C] = res[0].x[:12 * C]
    for r, fr in zip(res, parts):
        xg[12 * C:].reshape(F, 6)[fr] = r.x[12 * C:].reshape(-1, 6)
    assert np.abs(orc.predict_from_x(xg, C, p["obj"]) - orc.predict_from_x(whole.x, C, p["obj"])).max() < 1e-4
    if fixed:
        np.testing.assert_array_equal(xg[:12 * C].reshape(C, 12)[:, :6], x0[:12 * C].reshape(C, 12)[:, :6])


def test_frame_selection_matches_the_oracle_prefilter(mc, capsys):
    """The wrapper's pre-filter (bundle_adjustment.py:265-298) over random recordings: frames complete in two cameras, the 5 x nan-median
    threshold (or the caller's), the random subsample drawn from the global numpy RNG when n_frames is smaller than what is left --
    the same frames in the same order as the oracle, and the reference's printed line (threshold to 1e-12)."""
    rng = np.random.default_rng(77)
    for it in range(40 + EXTRA):
        C = int(rng.choice([2, 3, 5, 8]))
        F = int(rng.choice([1, 5, 63, 64, 65, 200, 333]))
        p = mc.synth.make_problem(C, F, rows=int(rng.integers(1, 4)), cols=int(rng.integers(2, 5)), seed=300 + it, missing=float(rng.choice([0.0, 0.2, 0.6])),
                                  scalar_nans=int(rng.choice([0, 5])), outlier_frames=int(rng.choice([0, 1, 3])) if F > 5 else 0)
        if rng.random() < 0.2:
            p["uvs"][int(rng.integers(C))] = np.nan                       # a camera that sees nothing
        thr = None if rng.random() < 0.6 else float(rng.choice([0.5, 5.0, 1e9]))
        nf = None if rng.random() < 0.4 else int(rng.integers(1, F + 3))
        args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
        tag = f"case {it}: C={C} F={F} n_frames={nf} threshold={thr}"
        np.random.seed(it)
        want, _, _, line = orc.prefilter_frames(*args, nf, thr)
        np.random.seed(it)
        got = mc.api.select_frames(*args, nf, thr)
        out = capsys.readouterr().out
        np.testing.assert_array_equal(got, want, err_msg=tag)
        assert got.dtype.kind == "i"
        # the printed line: the same counts; the threshold is 5 x the exact median of the GPU's own errors, which differ from numpy's in the last bits
        head, _, t_got = out.strip().splitlines()[-1].rpartition(" ")
        head_o, _, t_want = line.rpartition(" ")
        assert head == head_o, (tag, out, line)
        assert (t_got == t_want == "nan") or abs(float(t_got) - float(t_want)) <= 1e-12 * abs(float(t_want)), (tag, out, line)


@pytest.mark.parametrize("it", list(range(36)) + list(range(1000, 1000 + EXTRA)))
def test_bounded_runs_end_at_kkt_points_of_the_oracle_objective(mc, it):
    """bundle_adjust(..., bounds=(lo, hi)) on random problems with random boxes: bounds between the start and the UNCONSTRAINED optimum on a
    random tenth of all coordinates, on a quarter of the intrinsics, or on those + 14-23 board-pose coordinates (the optimum violates them); wide ones on a tenth of everything.  Judged with the
    oracle at the returned point: feasible, the cost between the unconstrained optimum's and the start's, scipy's active_mask, active
    coordinates exactly on their bounds, and the KKT conditions of the oracle's objective -- its gradient vanishes on the free coordinates
    (relative to the gradient at the start) and points outward on the active ones.  Sometimes with the intrinsics held fixed, with a
    callable loss, with a numeric x_scale."""
    from scipy.optimize._lsq.common import find_active_constraints

    sys_path_tests = os.path.dirname(os.path.abspath(__file__))
    import sys
    if sys_path_tests not in sys.path:
        sys.path.insert(0, sys_path_tests)
    from losses import charbonnier_quarter

    rng = np.random.default_rng(12000 + it)
    C = int(rng.choice([2, 3, 4, 6, 9, 14]))
    F = int(rng.integers(6, 30)) if C > 6 else int(rng.integers(8, 120))
    loss = [("soft_l1", 1.0), ("linear", 1.0), ("huber", 2.5), ("soft_l1", 0.5), (charbonnier_quarter, 0.7)][int(rng.integers(5))]
    mk = dict(n_cameras=C, n_frames=F, rows=int(rng.integers(2, 5)), cols=int(rng.integers(3, 6)), seed=1700 + it, perturb_seed=1800 + it,
              missing=float(rng.choice([0.0, 0.15])), scalar_nans=int(rng.choice([0, 5])))
    fixed = bool(rng.random() < 0.3)
    p = mc.synth.make_problem(**mk)
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    kw = dict(n_frames=None, ftol=1e-14, xtol=1e-14, gtol=1e-11, verbose=0, max_nfev=500, fix_intrinsics=fixed, loss=loss[0], f_scale=loss[1])
    tag = f"case {it}: {mk} loss={loss} fixed={fixed}"
    _, _, _, use, free = quiet(mc.bundle_adjust, *args, **kw)
    assert free.status > 0, tag
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use])
    xu, n, nc = free.x, x0.size, 12 * C
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    # Which coordinates get a bound the optimum violates: a tenth of ALL that moved (camera parameters and board poses alike -- boxes on
    # extrinsics / poses couple through the rig's gauge freedom: the projected steps of such runs bend, solver.py's working set handles it),
    # a quarter of the intrinsics (gauge-invariant: "k2 >= 0", "fx within 5 %"), or intrinsics + 14-23 pose coordinates close to the start
    moved = np.abs(xu - x0) > 1e-6 * np.maximum(1.0, np.abs(x0))
    if fixed:
        moved[:nc] &= (np.arange(nc) % 12) >= 6
    intr_c = np.nonzero(moved[:nc] & ((np.arange(nc) % 12) < 6))[0]
    pose_c = nc + np.nonzero(moved[nc:])[0]
    kind = it % 3 if not fixed else 2 * (it % 2)
    if kind == 0:
        cand = np.nonzero(moved)[0]
        pick = list(rng.choice(cand, size=max(2, cand.size // 10), replace=False))
        frac = {i: float(rng.uniform(0.2, 0.8)) for i in pick}
    else:
        pick = [] if fixed else list(rng.choice(intr_c, size=min(intr_c.size, max(2, intr_c.size // 4)), replace=False))
        frac = {i: float(rng.uniform(0.2, 0.8)) for i in pick}
        if kind == 2:
            extra = rng.choice(pose_c, size=min(pose_c.size, int(rng.integers(14, 24))), replace=False)
            pick += list(extra)
            frac.update({i: 0.1 for i in extra})
    for i in pick:
        b = x0[i] + frac[i] * (xu[i] - x0[i])
        if xu[i] > x0[i]:
            hi[i] = b
        else:
            lo[i] = b
    for i in rng.choice(n, size=max(2, n // 10), replace=False):   # ... and bounds nothing should touch (some on coordinates bounded above already)
        lo[i] = min(lo[i], min(x0[i], xu[i]) - 50.0) if np.isfinite(lo[i]) else min(x0[i], xu[i]) - 50.0
        hi[i] = hi[i] if np.isfinite(hi[i]) else max(x0[i], xu[i]) + 50.0
    assert np.all(lo < hi) and np.all(x0 >= lo) and np.all(x0 <= hi)
    xs = None
    if rng.random() < 0.25:   # least_squares' numeric x_scale (a fixed damping matrix instead of Marquardt's)
        xs = np.concatenate([np.tile([100.0, 100.0, 100.0, 100.0, 0.1, 0.1, 0.1, 0.1, 0.1, 10.0, 10.0, 10.0], C), np.tile([0.1, 0.1, 0.1, 10.0, 10.0, 10.0], use.size)])
    _, _, _, use2, res = quiet(mc.bundle_adjust, *args, **kw, bounds=(lo, hi), **({} if xs is None else dict(x_scale=xs)))
    np.testing.assert_array_equal(use2, use)
    # (status 0 = 500 evaluations spent: boxes on gauge-dependent coordinates can leave a nearly flat, curved valley to slide along, and ftol = xtol =
    #  1e-14 do not fire on gains of 1e-7 -- scipy's TRF creeps there as well.  Such a run must still HAVE reached a KKT point to the tolerance below.)
    assert res.status >= 0, (tag, res.status, res.nfev)
    x = res.x
    assert np.all(x >= lo) and np.all(x <= hi), tag
    uvs = p["uvs"][:, use]
    lname, fs = loss
    f = orc.residuals(x, uvs, p["obj"])
    cost = orc.robust_cost(f, lname, fs)
    c0 = orc.robust_cost(orc.residuals(x0, uvs, p["obj"]), lname, fs)
    assert abs(res.cost - cost) <= 1e-11 * cost + 1e-14, tag
    assert free.cost * (1 - 1e-9) <= cost <= c0 * (1 + 1e-12), (tag, free.cost, cost, c0)
    am = find_active_constraints(x, lo, hi, rtol=1e-14)
    np.testing.assert_array_equal(res.active_mask, am, err_msg=tag)
    np.testing.assert_allclose(x[am == -1], lo[am == -1], rtol=1e-14, atol=1e-14, err_msg=tag)   # (scipy's rule calls a coordinate within rtol of its bound active)
    np.testing.assert_allclose(x[am == 1], hi[am == 1], rtol=1e-14, atol=1e-14, err_msg=tag)
    # KKT with the oracle's gradient
    js, fsc = orc.robust_scales(f, lname, fs)
    J = orc.jacobian_csr(x, uvs, p["obj"])
    g = J.T @ (js * fsc)
    js0, fsc0 = orc.robust_scales(orc.residuals(x0, uvs, p["obj"]), lname, fs)
    g0 = np.abs(orc.jacobian_csr(x0, uvs, p["obj"]).T @ (js0 * fsc0))
    held = np.zeros(n, bool)
    if fixed:
        held[:nc] = (np.arange(nc) % 12) < 6
        g0 = g0[~held]
        np.testing.assert_array_equal(x[held], x0[held], err_msg=tag)
    scale = g0.max()
    off = (am == 0) & ~held
    kkt = 1e-6 if res.status > 0 else 1e-4   # (a run that spent its 500 evaluations sliding along a flat valley: close)
    assert np.abs(g[off]).max() <= kkt * scale + 1e-8, (tag, np.abs(g[off]).max(), scale, res.status, res.nfev)
    assert np.all(g[am == -1] >= -kkt * scale - 1e-8) and np.all(g[am == 1] <= kkt * scale + 1e-8), tag
    np.testing.assert_allclose(res.grad[~held], g[~held], rtol=0, atol=1e-6 * max(scale, np.abs(g).max()) + 1e-7, err_msg=tag)


@pytest.mark.parametrize("it", list(range(12)) + list(range(1000, 1000 + EXTRA)))
def test_callable_loss_runs_end_at_stationary_points_of_the_oracle_objective(mc, it):
    """least_squares' callable `loss` on random problems (1-20 cameras, boards of 4-20 points, missing detections, sometimes the intrinsics held
    fixed, f_scale 0.5-2.5): the point returned is a stationary point of the oracle's objective with the same function (scipy's own
    arithmetic), cost / fun / grad are the oracle's, and the table route agrees with the built-in name when the function IS soft_l1."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    from losses import charbonnier_quarter, soft_l1_as_callable

    mk, opts, fixed = draw(100 + it)
    fn = charbonnier_quarter if it % 3 else soft_l1_as_callable
    fs = float(np.random.default_rng(it).choice([0.5, 1.0, 2.5]))
    p = mc.synth.make_problem(**mk)
    C = mk["n_cameras"]
    args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    kw = dict(n_frames=None, ftol=1e-13, xtol=1e-13, gtol=1e-11, verbose=0, max_nfev=400, fix_intrinsics=fixed, loss=fn, f_scale=fs)
    tag = f"case {it}: {mk} {fn.__name__} f_scale={fs} fixed={fixed}"
    e, intr, poses, use, res = quiet(mc.bundle_adjust, *args, **kw)
    assert res.status > 0, tag
    if use.size == 0:
        return
    uvs = p["uvs"][:, use]
    x0 = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use])
    f = orc.residuals(res.x, uvs, p["obj"])
    cost = orc.robust_cost(f, fn, fs)
    assert abs(res.cost - cost) <= 1e-11 * cost + 1e-14, tag
    np.testing.assert_allclose(res.fun, f, rtol=0, atol=1e-9 * max(1.0, np.abs(f).max()), err_msg=tag)
    assert cost <= orc.robust_cost(orc.residuals(x0, uvs, p["obj"]), fn, fs) * (1 + 1e-12), tag
    js, fsc = orc.robust_scales(f, fn, fs)
    g = orc.jacobian_csr(res.x, uvs, p["obj"]).T @ (js * fsc)
    js0, fsc0 = orc.robust_scales(orc.residuals(x0, uvs, p["obj"]), fn, fs)
    g0 = np.abs(orc.jacobian_csr(x0, uvs, p["obj"]).T @ (js0 * fsc0))
    held = np.zeros(x0.size, bool)
    if fixed:
        held[:12 * C] = (np.arange(12 * C) % 12) < 6
        np.testing.assert_array_equal(res.x[held], x0[held], err_msg=tag)
    scale = g0[~held].max()
    assert np.abs(g[~held]).max() <= 1e-6 * scale + 1e-8, (tag, np.abs(g[~held]).max(), scale)
    np.testing.assert_allclose(res.grad[~held], g[~held], rtol=0, atol=1e-6 * max(scale, 1.0) + 1e-7, err_msg=tag)
    if fn is soft_l1_as_callable:   # the same optimum as the built-in name (device-resident loop): through what it predicts
        r2 = quiet(mc.bundle_adjust, *args, **dict(kw, loss="soft_l1"))[4]
        assert abs(r2.cost - res.cost) <= 1e-9 * res.cost + 1e-13, tag
        assert np.abs(orc.predict_from_x(r2.x, C, p["obj"]) - orc.predict_from_x(res.x, C, p["obj"])).max() < 1e-4, tag
