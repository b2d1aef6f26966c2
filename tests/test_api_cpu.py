"""Host logic of api.py that needs no GPU: the lazily materialised OptimizeResult, the sharded exact median (radix select over
all-reduced histograms), least_squares' numeric x_scale -- the LM driver runs on the CPU test double (tests/fake_problem.py)."""
import contextlib
import io
import pickle

import numpy as np
import pytest
from scipy.optimize import OptimizeResult

from fake_problem import OracleProblem
from multicam_calibration_amd import api, ops, solver, synth
from oracle import ba_oracle as orc


# ------------------------------------------------------------------ result.fun is produced on first access, by every route
def _lazy_result(counter):
    def thunk():
        counter.append(1)
        return np.arange(3.0)

    r = api.LazyOptimizeResult(OptimizeResult(x=np.zeros(2), cost=1.5, status=2))
    dict.__setitem__(r, "fun", api._Lazy(thunk))
    return r


@pytest.mark.parametrize("route", ["attr", "item", "get", "items", "values", "repr", "copy", "pickle"])
def test_lazy_result_materialises_once_whatever_the_route(route):
    n = []
    r = _lazy_result(n)
    assert "fun" in r and "fun" in r.keys() and "fun" in dir(r) and not n   # the field is listed without being produced
    assert r.cost == 1.5 and r["status"] == 2 and not n                      # other fields do not trigger it
    if route == "attr":
        v = r.fun
    elif route == "item":
        v = r["fun"]
    elif route == "get":
        v = r.get("fun")
    elif route == "items":
        v = dict(r.items())["fun"]
    elif route == "values":
        v = [u for u in r.values() if isinstance(u, np.ndarray) and u.size == 3][0]
    elif route == "repr":
        assert "fun: [" in repr(r)
        v = r.fun
    elif route == "copy":
        c = r.copy()
        assert type(c) is OptimizeResult
        v = c.fun
    else:
        c = pickle.loads(pickle.dumps(r))
        assert type(c) is OptimizeResult
        v = c.fun
    np.testing.assert_array_equal(v, np.arange(3.0))
    np.testing.assert_array_equal(r.fun, np.arange(3.0))
    assert len(n) == 1
    assert not isinstance(dict.get(r, "fun"), api._Lazy)


# ------------------------------------------------------------------ exact median from sharded histograms == np.nanmedian
@pytest.mark.parametrize("shards", [1, 2, 5])
@pytest.mark.parametrize("seed,missing", [(0, 0.0), (1, 0.3), (2, 0.6)])
def test_sharded_radix_select_equals_nanmedian(shards, seed, missing):
    p = synth.make_problem(3, 37, seed=seed, missing=missing)
    x = api.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    bounds = api._split_bounds(37, shards)
    rng = np.random.default_rng(seed)
    mask = rng.uniform(size=37) < 0.8
    probs = []
    for r in range(shards):
        lo, hi = bounds[r], bounds[r + 1]
        q = OracleProblem(p["uvs"][:, lo:hi], p["obj"])
        q.set_params(0, np.concatenate([x[:36], x[36 + 6 * lo:36 + 6 * hi]]))
        q.frame_errors(0)
        probs.append((q, mask[lo:hi]))
    first = {"v": True}

    def hist(prefix, pas):   # what the all-reduce of the per-rank histograms yields
        h = sum(q.error_histogram(m if first["v"] else None, prefix, pas).astype(np.int64) for q, m in probs)
        first["v"] = False
        return h

    whole = OracleProblem(p["uvs"], p["obj"])
    whole.set_params(0, x)
    whole.frame_errors(0)
    want = whole.error_median(mask)[0]
    got = api._median_from_histograms(hist)
    assert got == want or (np.isnan(got) and np.isnan(want))


def test_sharded_radix_select_of_nothing_is_nan():
    assert np.isnan(api._median_from_histograms(lambda prefix, p: np.zeros(256, dtype=np.int64)))


def test_split_bounds_are_array_split_sizes():
    for n in (0, 1, 7, 41, 100000):
        for w in (1, 2, 3, 8):
            want = np.concatenate([[0], np.cumsum([len(a) for a in np.array_split(np.arange(n), w)])])
            np.testing.assert_array_equal(api._split_bounds(n, w), want)


# ------------------------------------------------------------------ least_squares' numeric x_scale
def test_x_scale_validation_is_scipys():
    assert api._check_x_scale("jac", 10) is None
    np.testing.assert_array_equal(api._check_x_scale(2.0, 4), np.full(4, 2.0))
    for bad in (0.0, -1.0, np.array([1.0, np.nan]), "foo", np.array([1.0, -2.0])):
        with pytest.raises(ValueError, match="`x_scale` must be 'jac' or array_like with positive numbers."):
            api._check_x_scale(bad, 2)
    with pytest.raises(ValueError, match="Inconsistent shapes between `x_scale` and `x0`."):
        api._check_x_scale(np.ones(3), 4)


def test_numeric_x_scale_changes_the_path_not_the_minimiser(monkeypatch):
    """D = 1 / x_scale^2 instead of diag(J^T J): another damping, hence other iterates, the same stationary point."""
    monkeypatch.setattr(ops, "Problem", OracleProblem)
    p = synth.make_problem(2, 14, seed=5)
    n_total = 24 + 6 * 14
    xs = np.concatenate([np.tile([100.0, 100.0, 50.0, 50.0, 0.05, 0.05, 0.01, 0.01, 0.01, 5.0, 5.0, 5.0], 2), np.tile([0.01, 0.01, 0.01, 5.0, 5.0, 5.0], 14)])
    kw = dict(n_frames=None, verbose=0, return_jac=False, ftol=0.0, xtol=1e-13, gtol=1e-9, max_nfev=200, outlier_threshold=1e9)
    with contextlib.redirect_stdout(io.StringIO()):
        a = api.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)[4]
        b = api.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], x_scale=xs, **kw)[4]
        c = api.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], x_scale=1.0, **kw)[4]
    assert xs.size == n_total
    assert a.status > 0 and b.status > 0 and c.status > 0
    ha, hb = [h[2] for h in a.lm["history"]], [h[2] for h in b.lm["history"]]
    assert ha[:3] != hb[:3]                                    # other trial costs from the first step on
    for r in (b, c):
        assert abs(r.cost - a.cost) <= 1e-9 * a.cost
        pa, pb = orc.predict_from_x(a.x, 2, p["obj"]), orc.predict_from_x(r.x, 2, p["obj"])
        assert np.abs(pa - pb).max() < 1e-5
    with pytest.raises(ValueError, match="Inconsistent shapes"):
        with contextlib.redirect_stdout(io.StringIO()):
            api.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], x_scale=np.ones(5), **kw)


def test_result_records_which_backend_ran(monkeypatch):
    monkeypatch.setattr(ops, "Problem", OracleProblem)
    p = synth.make_problem(2, 8, seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        res = api.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0, return_jac=False)[4]
    assert res.lm["collectives"] == "SingleProcess" and res.lm["world"] == 1 and res.lm["reduced_solver"] == "host"
    f = orc.residuals(res.x, p["uvs"], p["obj"])
    np.testing.assert_allclose(res.fun, f, rtol=0, atol=1e-9)   # the lazily attached residual vector, NaN scalars removed
