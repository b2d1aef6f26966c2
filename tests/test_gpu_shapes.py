"""GPU parity on the regime boundaries of the kernels (run with `-m gpu` on an MI355X): camera counts either side of every
variant switch (factor of the reduced system in LDS <= 9 cameras, 256 / 512 / 1024-thread k_solve_cam, k_syrk's stage
size 8 / 4 / 2 frames, its 4- and 16-tile wavefronts), frame counts around the 64-frame wave tile, boards of 1 .. 7 points.
Each shape: normal equations, Schur reduction, frame gradient, cost and one device-solved camera step against the oracle."""
import numpy as np
import pytest

from oracle import ba_oracle as orc

pytestmark = pytest.mark.gpu

FRAMES_BOARDS = ((1, (2, 2)), (2, (1, 3)), (63, (1, 1)), (64, (1, 2)), (65, (2, 2)), (130, (1, 5)), (7, (1, 7)))


@pytest.fixture(scope="module")
def mc():
    import multicam_calibration_amd as m

    m.ops.load_library()
    return m


@pytest.mark.parametrize("C", [1, 2, 5, 8, 9, 10, 16, 17, 26, 27, 32, 33])
def test_regime_boundaries_vs_oracle(mc, C):
    for F, (rows, cols) in FRAMES_BOARDS:
        p = mc.synth.make_problem(C, F, rows=rows, cols=cols, pitch=50.0, seed=100 + C + F, missing=0.2 if F > 2 else 0.0)
        tag = f"C={C} F={F} N={rows * cols}"
        x = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
        lam = 1e-2
        prob = mc.ops.Problem(p["uvs"], p["obj"])
        prob.set_params(0, x)
        prob.linearize(0)
        prob.build_reduced(lam)
        red = {k: v.copy() for k, v in prob.get_reduced().items()}
        gfd = prob.frame_gradient()
        U, gc, V, gf, W, cost = orc.normal_equations(x, p["uvs"], p["obj"])
        Df2 = np.stack([np.where(np.diag(V[f]) > 0, np.diag(V[f]), 1.0) for f in range(F)])
        S, rhs = orc.schur_reduce(U, gc, V, gf, W, lam, np.zeros((C, 12)), Df2)
        assert np.abs(red["S0"] - S).max() <= 1e-10 * np.abs(S).max(), tag
        assert np.abs(red["rhs"] - rhs).max() <= 1e-10 * np.abs(rhs).max(), tag
        assert abs(red["scal"][0] - cost) <= 1e-12 * cost, tag
        assert np.abs(gfd - gf).max() <= 1e-10 * np.abs(gf).max(), tag
        prob.lm_set_state(float(red["scal"][0]), lam, 2.0, 0)
        prob.lm_auto_config(0.0, 0.0, 0.0, 1e-12, 1e12, None)
        prob.lm_auto_solve(1)
        st = prob.lm_auto_wait(1).copy()
        dc = prob.cam_step()
        prob.close()
        assert st[23] == 0, tag                                                 # the damped system is positive definite
        Sd = red["S0"] + lam * np.diag(np.where(red["diagU"] > 0, red["diagU"], 1.0))
        ref = np.linalg.solve(Sd, red["rhs"])
        assert np.abs(dc - ref).max() <= 1e-6 * np.abs(ref).max(), tag


def _with_env(env, fn):
    import os

    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_launch_variants_agree_across_the_item_count_regimes(mc):
    """Round 4: which k_gram launch runs depends on items = cameras x frame blocks (<= 256: point split by 4, <= 512: by 2, <= 1 024:
    fused, beyond: fused rounds + a point-split / split-role / fused tail in one or two launches) and on the camera block width.  Thirty
    random shapes across those regimes -- sizes the oracle cannot do densely -- each linearised with the library's own choice and with the
    plain fused kernel forced (MCBA_GRAM_SPLIT=0): reduced system, right-hand side, cost and frame gradients must agree to round-off
    (different summation orders over the points, nothing else); the 6-wide camera block must equal rows 6..11 of the 12-wide system."""
    rng = np.random.default_rng(2024)
    lam = 3e-3
    for it in range(30):
        C = int(rng.integers(1, 13))
        regime = it % 5
        items = [int(rng.integers(1, 200)), int(rng.integers(257, 500)), int(rng.integers(520, 1000)), int(rng.integers(1030, 1250)), int(rng.integers(1300, 2300))][regime]
        nfb = max(1, items // C)
        F = int(64 * nfb - rng.integers(0, 64))
        rows, cols = int(rng.integers(1, 4)), int(rng.integers(1, 8))
        p = mc.synth.make_problem(C, F, rows=rows, cols=cols, pitch=40.0, seed=500 + it, missing=float(rng.choice([0.0, 0.15])))
        tag = f"C={C} F={F} N={rows * cols} items={C * ((F + 63) // 64)}"
        x = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])

        def linearised(cw):
            prob = mc.ops.Problem(p["uvs"], p["obj"])
            if cw == 6:
                assert prob.set_camera_block(6)
            prob.set_params(0, x)
            prob.linearize(0)
            prob.build_reduced(lam)
            red = {k: v.copy() for k, v in prob.get_reduced().items()}
            gfd = prob.frame_gradient().copy()
            keep = np.asarray(prob.cam_index)
            prob.close()
            return red, gfd, keep

        ref, gref, _ = _with_env({"MCBA_GRAM_SPLIT": "0"}, lambda: linearised(12))
        got, ggot, _ = linearised(12)
        six, gsix, keep = linearised(6)
        scale = np.abs(ref["S0"]).max()
        for a, b, what in ((got["S0"], ref["S0"], "S0"), (six["S0"], ref["S0"][np.ix_(keep, keep)], "S0 (6-wide)")):
            assert np.abs(a - b).max() <= 1e-11 * scale, (tag, what)
        assert np.abs(got["rhs"] - ref["rhs"]).max() <= 1e-11 * np.abs(ref["rhs"]).max(), tag
        assert np.abs(six["rhs"] - ref["rhs"][keep]).max() <= 1e-11 * np.abs(ref["rhs"]).max(), tag
        assert abs(got["scal"][0] - ref["scal"][0]) <= 1e-13 * ref["scal"][0] and abs(six["scal"][0] - ref["scal"][0]) <= 1e-13 * ref["scal"][0], tag
        assert got["scal"][1] == ref["scal"][1] == six["scal"][1], tag
        assert np.abs(ggot - gref).max() <= 1e-11 * np.abs(gref).max() and np.abs(gsix - gref).max() <= 1e-11 * np.abs(gref).max(), tag
