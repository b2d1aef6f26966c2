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


@pytest.mark.parametrize("C", [2, 3, 6, 8, 9, 24, 40])   # <= 8: one lane per point; beyond: one wavefront per point, pairs across the lanes
def test_matches_oracle_with_noise_and_missing_views(mc, C):
    uvs, ext, intr, X = scene(C=C, P=1000 if C <= 8 else 300, seed=10 + C, noise=0.3, p_unseen=0.25 if C <= 8 else 0.6)
    want = tri.triangulate(uvs, ext, intr)
    got = mc.triangulate(uvs, ext, intr)
    assert got.shape == want.shape
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
    with pytest.raises(NotImplementedError):
        mc.triangulate(uvs * 22, np.tile(ext, (22, 1)), intr * 22)   # 66 cameras
    empty = mc.triangulate([u[:0] for u in uvs], ext, intr)
    assert empty.shape == (0, 3)


def test_the_config5_rig_triangulates(mc):
    """24 cameras (BASELINE configs[4]'s rig): 276 camera pairs per point; noise-free points are recovered, points seen by
    fewer than two cameras come back NaN, and the full 5-coefficient distortion model is honoured."""
    uvs, ext, intr, X = scene(C=24, P=500, seed=77)
    assert np.abs(mc.triangulate(uvs, ext, intr, undistort_iterations=20) - X).max() < 1e-8
    uvs2 = [u.copy() for u in uvs]
    for c in range(1, 24):
        uvs2[c][:7] = np.nan               # points 0..6: seen by camera 0 only
    for c in range(24):
        uvs2[c][7:9] = np.nan              # points 7, 8: seen by nobody
    got = mc.triangulate(uvs2, ext, intr, undistort_iterations=20)
    assert np.isnan(got[:9]).all() and np.abs(got[9:] - X[9:]).max() < 1e-8
    intr_t = [(K, np.array([d[0], d[1], 1e-3, -5e-4, 1e-3])) for K, d in intr]
    want = tri.triangulate(uvs, ext, intr_t)
    assert np.abs(mc.triangulate(uvs, ext, intr_t) - want).max() <= 1e-8 * np.abs(want).max()


def test_random_rigs_and_occlusion_patterns_vs_oracle(mc):
    """Seeded random sweep: 2..64 cameras (both kernels, the 8 | 9 switch, the 64-camera limit), point counts around the 64-lane
    wavefront, occlusion from none to almost everything (points left with 0, 1, 2 views), single missing coordinates -- NaN pattern
    identical to the oracle's, values to the DLT's conditioning."""
    rng = np.random.default_rng(404)
    for it in range(40):
        C = int(rng.choice([2, 3, 4, 7, 8, 9, 10, 16, 33, 63, 64]))
        P = int(rng.choice([1, 2, 63, 64, 65, 130, 257]))
        uvs, ext, intr, X = scene(C=C, P=P, seed=1000 + it, noise=float(rng.choice([0.0, 0.3])), p_unseen=float(rng.choice([0.0, 0.3, 0.7, 0.95])))
        uvs = [u.copy() for u in uvs]
        for _ in range(int(rng.integers(0, 4))):           # a detection with ONE coordinate missing counts as unseen (geometry.py: isnan(...).any)
            uvs[int(rng.integers(C))][int(rng.integers(P)), int(rng.integers(2))] = np.nan
        want = tri.triangulate(uvs, ext, intr)
        got = mc.triangulate(uvs, ext, intr)
        tag = f"case {it}: C={C} P={P}"
        assert got.shape == want.shape == (P, 3), tag
        assert np.array_equal(np.isnan(got), np.isnan(want)), tag
        ok = ~np.isnan(want).any(1)
        if ok.any():
            assert np.abs(got[ok] - want[ok]).max() <= 1e-7 * max(1.0, np.abs(want[ok]).max()), (tag, np.abs(got[ok] - want[ok]).max())
