"""Pin the CPU oracle (oracle/ba_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import contextlib
import io

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ba_oracle as orc
from conftest import problem_from_npz


@pytest.mark.parametrize("name", ["complete", "missing", "fourcam", "edge"])
def test_residuals_match_reference(golden, name):
    z = golden("residuals.npz")
    uvs, ext, intr, obj, poses = problem_from_npz(z, name + "_")
    x0 = orc.serialize_params(ext, intr, poses)
    np.testing.assert_array_equal(x0, z[name + "_x0"])
    r = orc.residuals(x0, uvs, obj)
    assert r.shape == z[name + "_res"].shape
    # closed form vs the reference's 4x4 homogeneous chain: round-off only (<= 3.4e-13 px)
    np.testing.assert_allclose(r, z[name + "_res"], rtol=0, atol=1e-11)


def test_rodrigues_theta_zero_convention(golden):
    z = golden("residuals.npz")
    R = orc.rodrigues(z["rodrigues_in"])
    np.testing.assert_allclose(R, z["rodrigues_out"], rtol=0, atol=1e-15)
    assert np.array_equal(R[0], np.eye(3))


def test_deserialize_roundtrip_drops_p1p2k3():
    intr = [(np.array([[1000.0, 0, 640], [0, 1010, 512], [0, 0, 1]]), np.array([-0.1, 0.02, 0.3, 0.4, 0.5]))]
    x = orc.serialize_params(np.zeros((1, 6)), intr, np.ones((2, 6)))
    assert x.shape == (12 + 12,)
    ext, intr2, poses = orc.deserialize_params(x, 1)
    np.testing.assert_array_equal(intr2[0][1], [-0.1, 0.02, 0, 0, 0])
    np.testing.assert_array_equal(intr2[0][0], intr[0][0])


def test_sparsity_matches_reference(golden):
    z = golden("sparsity.npz")
    A = orc.sparsity_csr(z["uvs"])
    A.sort_indices()
    assert tuple(z["shape"]) == A.shape
    np.testing.assert_array_equal(A.indptr, z["indptr"])
    np.testing.assert_array_equal(A.indices, z["indices"])


def test_analytic_jacobian_vs_reference_fd(golden):
    z = golden("jacobian.npz")
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    x0 = orc.serialize_params(ext, intr, poses)
    J = orc.jacobian_csr(x0, uvs, obj)
    J.sort_indices()
    np.testing.assert_array_equal(J.indices, z["J3_indices"])
    np.testing.assert_array_equal(J.indptr, z["J3_indptr"])
    J3 = sp.csr_matrix((z["J3_data"], z["J3_indices"], z["J3_indptr"]), shape=J.shape)
    J2 = sp.csr_matrix((z["J2_data"], z["J2_indices"], z["J2_indptr"]), shape=J.shape)
    rel3 = sp.linalg.norm(J - J3) / sp.linalg.norm(J3)
    rel2 = sp.linalg.norm(J - J2) / sp.linalg.norm(J2)
    assert rel3 < 1e-8, rel3  # 5.7e-10 vs 3-point FD
    assert rel2 < 2e-7, rel2  # 2.5e-8 vs the reference's own 2-point FD
    # the theta = 0 root camera's rotation columns (6..8) must be as good as the rest
    cols = slice(6, 9)
    d = np.abs((J - J3).toarray()[:, cols]).max() / np.abs(J3.toarray()[:, cols]).max()
    assert d < 1e-8, d
    assert int(z["n_groups"]) == 18


def test_prefilter_matches_reference(golden):
    z = golden("prefilter.npz")
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    for i in range(int(z["n_cases"])):
        n_frames, seed, thr = z[f"case{i}_args"]
        n_frames = None if n_frames < 0 else int(n_frames)
        thr = None if thr < 0 else float(thr)
        np.random.seed(int(seed))
        use, _, _, line = orc.prefilter_frames(uvs, ext, intr, obj, poses, n_frames, thr)
        np.testing.assert_array_equal(use, z[f"case{i}_use"])
        # identical text; the threshold it prints is 5x a nan-median of round-off-level-different errors
        a, b = line.rsplit(" ", 1), str(z[f"case{i}_line"]).rsplit(" ", 1)
        assert a[0] == b[0]
        assert abs(float(a[1]) - float(b[1])) <= 1e-9 * abs(float(b[1]))
        # same consumption of the GLOBAL numpy RNG as the reference
        assert np.random.randint(0, 2**31 - 1) == int(z[f"case{i}_rng_after"])


@pytest.mark.parametrize("loss", ["soft_l1", "huber", "cauchy", "arctan"])
@pytest.mark.parametrize("fs", [1.0, 2.5])
def test_robust_loss_matches_scipy(golden, loss, fs):
    z = golden("robust.npz")
    f, J = z["f"], z["J"]
    rho = z[f"{loss}_{fs}_rho"]
    r0, r1, r2 = orc.loss_rho((f / fs) ** 2, loss)
    np.testing.assert_allclose(r0 * fs**2, rho[0], rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(r1, rho[1], rtol=1e-14)
    np.testing.assert_allclose(r2 / fs**2, rho[2], rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(orc.robust_cost(f, loss, fs), float(z[f"{loss}_{fs}_cost"]), rtol=1e-14)
    js, fsc = orc.robust_scales(f, loss, fs)
    np.testing.assert_allclose(J * js[:, None], z[f"{loss}_{fs}_J"], rtol=1e-13)
    np.testing.assert_allclose(fsc, z[f"{loss}_{fs}_f"], rtol=1e-13)


def test_default_path_reproduces_reference_run(golden):
    """Same third-party least_squares call on the restated residual -> same iteration counts and cost."""
    z = golden("default_run.npz")
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        e, i, p, use, res = orc.bundle_adjust(uvs, ext, intr, obj, poses, n_frames=None)
    assert (res.nfev, res.njev, res.status) == (int(z["nfev"]), int(z["njev"]), int(z["status"]))
    assert abs(res.cost - float(z["cost"])) <= 1e-6 * float(z["cost"])
    np.testing.assert_array_equal(use, z["use"])
    # round-off sensitive trajectory: parameters agree loosely, predictions tightly
    pred_a = orc.predict_from_x(res.x, uvs.shape[0], obj)
    pred_b = orc.predict_from_x(z["x"], uvs.shape[0], obj)
    assert np.abs(pred_a - pred_b).max() < 5e-2
    assert buf.getvalue().splitlines()[0].rsplit(" ", 1)[0] == str(z["log"]).splitlines()[0].rsplit(" ", 1)[0]
