"""triangulate() on the GPU (k_triangulate through the C ABI) against the numpy oracle (parity unpinned: OpenCV is
absent; the oracle restates its two published algorithms) and against synthetic truth."""
import numpy as np
import pytest

from oracle import triangulate_oracle as tri
from test_triangulate_cpu import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc():
    import multicam_calibration_amd as m

    m.ops.load_library()
    return m


@pytest.mark.parametrize("C", [2, 3, 6, 8])
def test_matches_oracle_with_noise_and_missing_views(mc, C):
    uvs, ext, intr, X = scene(C=C, P=1000, seed=10 + C, noise=0.3, p_unseen=0.25)
    want = tri.triangulate(uvs, ext, intr)
    got = mc.triangulate(uvs, ext, intr)
    assert got.shape == (1000, 3)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want).any(1)
    # same algorithm in FP64: SVD by LAPACK vs one-sided Jacobi differ in the last bits, scaled by the DLT conditioning
    assert np.abs(got[ok] - want[ok]).max() <= 1e-8 * np.abs(want[ok]).max()


def test_recovers_noise_free_points_and_tangential_distortion(mc):
    uvs, ext, intr, X = scene(C=5, P=777, seed=21)
    assert np.abs(mc.triangulate(uvs, ext, intr, undistort_iterations=20) - X).max() < 1e-8
    intr_t = [(K, np.array([d[0], d[1], 1e-3, -5e-4, 1e-3])) for K, d in intr]   # p1, p2, k3: full 5-coefficient model
    want = tri.triangulate(uvs, ext, intr_t)
    got = mc.triangulate(uvs, ext, intr_t)
    assert np.abs(got - want).max() <= 1e-8 * np.abs(want).max()


def test_argument_checks(mc):
    uvs, ext, intr, X = scene(C=3, P=10)
    with pytest.raises(ValueError):
        mc.triangulate(uvs[:2], ext, intr)
    with pytest.raises(NotImplementedError):
        mc.triangulate(uvs[:1], ext[:1], intr[:1])
    empty = mc.triangulate([u[:0] for u in uvs], ext, intr)
    assert empty.shape == (0, 3)
