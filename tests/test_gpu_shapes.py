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
