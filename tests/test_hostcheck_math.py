"""csrc/mcba_math.h (the text compiled into the HIP kernels) checked on the CPU against the oracle.
Runs without a GPU: tests/hostcheck/hostcheck.cpp wraps the header for g++."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import ba_oracle as orc
from multicam_calibration_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
dp = ctypes.POINTER(ctypes.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def hc():
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    lib = os.path.join(HERE, "hostcheck", "libhostcheck.so")
    hdrs = [os.path.join(HERE, "..", "multicam-calibration_amd", "csrc", h) for h in ("mcba_math.h", "mcba_lm.h", "mcba_lm_state.h", "mcba_pnp_math.h")]
    flags = ["-O2"]
    if os.environ.get("MCBA_HOSTCHECK_SANITIZE") == "1":   # the run test_hostcheck_under_sanitizers starts: AddressSanitizer + UBSan build of the same text
        lib = os.path.join(HERE, "hostcheck", "libhostcheck_san.so")
        flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++"] + flags + ["-shared", "-fPIC", "-o", lib, src])
    return ctypes.CDLL(lib)


def tri_to_full(t, n):
    out = np.zeros(t.shape[:-1] + (n, n))
    iu = np.triu_indices(n)
    out[..., iu[0], iu[1]] = t
    out[..., iu[1], iu[0]] = t
    return out


def drot_mp(r, X, h=None):
    """d(R(r) X)/dr by 60-digit central differences of the reference's Rodrigues formula (mpmath)."""
    import mpmath as mp

    mp.mp.dps = 60

    def RX(rv):
        th = mp.sqrt(sum(v * v for v in rv))
        K = mp.matrix([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]])
        if th == 0:
            return mp.matrix(X)
        A = K / th
        return (mp.eye(3) + mp.sin(th) * A + (1 - mp.cos(th)) * A * A) * mp.matrix(X)

    h = mp.mpf(10) ** -25
    G = np.zeros((3, 3))
    for k in range(3):
        rp = [mp.mpf(float(v)) for v in r]
        rm = list(rp)
        rp[k] += h
        rm[k] -= h
        d = (RX(rp) - RX(rm)) / (2 * h)
        G[:, k] = [float(d[i]) for i in range(3)]
    return G


def test_rotation_constants(hc):
    for r in [np.zeros(3), np.array([1e-9, -2e-9, 3e-9]), np.array([3e-3, 4e-3, -1e-3]), np.array([0.0099, 0, 0]), np.array([0.0101, 0, 0]), np.array([0.3, -0.2, 0.1]), np.array([2.0, 1.5, -1.0])]:
        R, Jr = np.zeros(9), np.zeros(9)
        hc.hc_rot(P(r), P(R), P(Jr))
        np.testing.assert_allclose(R.reshape(3, 3), orc.rodrigues(r), rtol=0, atol=2e-16 * 4)
        # right Jacobian: d(R(r) X)/dr = -R [X]x Jr   vs the oracle's closed form G(r, X)
        X = np.array([0.7, -1.3, 2.1])
        G = -(R.reshape(3, 3) @ orc.skew(X)) @ Jr.reshape(3, 3)
        np.testing.assert_allclose(G, orc.drot_point(r, X), rtol=0, atol=1e-13)
        np.testing.assert_allclose(G, drot_mp(r, X), rtol=0, atol=1e-13)


@pytest.mark.parametrize("kw", [dict(n_cameras=2, n_frames=5, seed=3), dict(n_cameras=3, n_frames=7, seed=4, missing=0.3, scalar_nans=9), dict(n_cameras=4, n_frames=3, seed=5, rows=5, cols=7)])
def test_jacobian_rows_match_oracle(hc, kw):
    p = synth.make_problem(**kw)
    C, F, N, _ = p["uvs"].shape
    x = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    pred, Jc, Jf = np.zeros((C, F, N, 2)), np.zeros((C, F, N, 2, 12)), np.zeros((C, F, N, 2, 6))
    hc.hc_jac_rows(C, F, N, P(p["obj"]), P(x), P(pred), P(Jc), P(Jf))
    np.testing.assert_allclose(pred, orc.predict_from_x(x, C, p["obj"]), rtol=0, atol=1e-10)
    Jc0, Jf0 = orc.jacobian_blocks(x, C, p["obj"])
    assert np.abs(Jc - Jc0).max() <= 1e-11 * np.abs(Jc0).max()
    assert np.abs(Jf - Jf0).max() <= 1e-11 * np.abs(Jf0).max()
    # per-column scale check too (columns differ by orders of magnitude)
    for k in range(12):
        assert np.abs(Jc[..., k] - Jc0[..., k]).max() <= 1e-10 * max(np.abs(Jc0[..., k]).max(), 1e-300)


@pytest.mark.parametrize("loss,fs", [("soft_l1", 1.0), ("linear", 1.0), ("huber", 0.4), ("cauchy", 2.0), ("arctan", 1.5)])
def test_normal_equations_match_oracle(hc, loss, fs):
    p = synth.make_problem(3, 6, seed=6, missing=0.25, scalar_nans=5)
    C, F, N, _ = p["uvs"].shape
    x = orc.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    U, gc = np.zeros((C, 78)), np.zeros((C, 12))
    W, V, gf = np.zeros((C, F, 72)), np.zeros((C, F, 21)), np.zeros((C, F, 6))
    cost = np.zeros(1)
    lid = ["linear", "soft_l1", "huber", "cauchy", "arctan"].index(loss)
    hc.hc_normal_eq(C, F, N, P(p["uvs"]), P(p["obj"]), P(x), lid, ctypes.c_double(fs), P(U), P(gc), P(W), P(V), P(gf), P(cost))
    U0, gc0, V0, gf0, W0, cost0 = orc.normal_equations(x, p["uvs"], p["obj"], loss, fs)
    assert abs(cost[0] - cost0) <= 1e-12 * cost0
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(tri_to_full(U, 12), U0) < 1e-12
    assert rel(gc, gc0) < 1e-11
    assert rel(W.reshape(C, F, 12, 6), W0) < 1e-12
    assert rel(tri_to_full(V, 6).sum(0), V0) < 1e-12
    assert rel(gf.sum(0), gf0) < 1e-11


def test_chol6(hc):
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 9))
    V = A @ A.T
    b = rng.normal(size=6)
    x = np.zeros(6)
    ok = hc.hc_chol_solve(P(V[np.triu_indices(6)].copy()), P(b), P(x))
    assert ok == 1
    np.testing.assert_allclose(x, np.linalg.solve(V, b), rtol=1e-11)


# ---- calibrate()'s per-view arithmetic (csrc/mcba_pnp_math.h: the text of k_pnp) against oracle/calibration_oracle.py, on the CPU
def _pnp_views(seed, noise, n_views=14):
    p = synth.make_problem(2, n_views, seed=seed, noise=noise)
    cam = p["true_cam"][1]
    return p["obj"], p["uvs"][1], np.r_[cam[:6], 0.0, 0.0, 0.0]


@pytest.mark.parametrize("noise", [0.0, 0.2, 3.0])
def test_view_homography_is_the_numpy_dlt(hc, noise):
    """One lane's homography (Hartley-normalised DLT; smallest eigenvector of the 9 x 9 normal matrix by inverse iteration on a block Cholesky
    factor) = numpy's SVD null vector to 1e-9, in pixel coordinates (cv2.calibrateCamera's start) and in undistorted normalised ones
    (cv2.solvePnP's); an incomplete view is reported as such."""
    from oracle import calibration_oracle as co

    obj, uvs, intr9 = _pnp_views(80, noise)
    K = np.array([[intr9[0], 0, intr9[2]], [0, intr9[1], intr9[3]], [0, 0, 1.0]])
    hc.hc_view_homography.restype = ctypes.c_int
    for v, uv in enumerate(uvs):
        H = np.zeros(9)
        assert hc.hc_view_homography(len(obj), P(np.ascontiguousarray(uv)), P(obj), None, 0, P(H)) == 1
        want = co.homographies(obj[:, :2], uv[None])[0].ravel()
        assert (np.abs(H - want) / np.maximum(1.0, np.abs(want))).max() < 1e-9 and H[8] == 1.0, v
        assert hc.hc_view_homography(len(obj), P(np.ascontiguousarray(uv)), P(obj), P(intr9), 8, P(H)) == 1
        want = co.homographies(obj[:, :2], co.undistort_normalized(uv[None], K, intr9[4:]))[0].ravel()
        assert (np.abs(H - want) / np.maximum(1.0, np.abs(want))).max() < 1e-9, v
    gone = uvs[0].copy()
    gone[7, 1] = np.nan
    assert hc.hc_view_homography(len(obj), P(gone), P(obj), None, 0, P(np.zeros(9))) == 0


@pytest.mark.parametrize("five", [False, True])
def test_view_pose_is_the_reprojection_minimiser(hc, five):
    """One lane's cv2.solvePnP: the start = numpy's pose from the homography (polar factor by SVD, the reference's rodrigues_inv) to 1e-8, the
    result = scipy's minimiser of the pixel reprojection error to 1e-9 in cost (two- and five-coefficient intrinsics), the truth on noise-free
    detections, a handful of linearisations."""
    from oracle import calibration_oracle as co

    hc.hc_view_pose.restype = ctypes.c_int
    intr_extra = np.array([1.5e-3, -8e-4, -0.03]) if five else np.zeros(3)
    rng = np.random.default_rng(3)
    obj = synth.board_points()
    intr9 = np.r_[1180.0, 1170.0, 650.0, 505.0, -0.09, 0.03, intr_extra]
    K = np.array([[intr9[0], 0, intr9[2]], [0, intr9[1], intr9[3]], [0, 0, 1.0]])
    for v in range(12):
        pose_true = np.r_[rng.normal(0, 0.4, 3), rng.normal(0, 40, 2) - 40.0, rng.uniform(450, 800)]
        for noise in (0.0, 0.3):
            uv = co.project5(obj, pose_true, intr9) + rng.normal(0, noise, (len(obj), 2))
            start, pose, cost = np.zeros(6), np.zeros(6), np.zeros(1)
            n = hc.hc_view_pose(len(obj), P(np.ascontiguousarray(uv)), P(obj), P(intr9), 8, 60, P(start), P(pose), P(cost))
            assert 1 <= n <= 25, (v, noise, n)   # (noise-free: the homography start IS the minimiser -- one linearisation sees a zero gradient)
            want0 = co.poses_from_homographies(co.homographies(obj[:, :2], co.undistort_normalized(uv[None], K, intr9[4:])), np.eye(3))[0]
            np.testing.assert_allclose(start, want0, rtol=0, atol=1e-8 * max(1.0, np.abs(want0).max()))
            mine = 0.5 * np.sum((uv - co.project5(obj, pose, intr9)) ** 2)
            assert abs(mine - cost[0]) <= 1e-9 * max(mine, 1e-20) + 1e-22
            if noise == 0.0:
                assert np.abs(co.project5(obj, pose, intr9) - uv).max() < 1e-8
            else:
                _, c_ref = co.solve_pnp(uv, obj, intr9, want0)
                assert abs(mine - c_ref) <= 1e-9 * c_ref, (v, mine, c_ref)
    # an incomplete view: no pose is attempted
    uv[3, 0] = np.nan
    assert hc.hc_view_pose(len(obj), P(np.ascontiguousarray(uv)), P(obj), P(intr9), 8, 60, P(np.zeros(6)), P(np.zeros(6)), P(np.zeros(1))) == 0


def test_zhang_closed_form_is_the_numpy_null_vector(hc):
    """k_zhang's arithmetic for one camera (normal matrix of Zhang's rows view by view, smallest eigenvector by cyclic Jacobi, the closed form) =
    the numpy restatement's K (last right singular vector of the stacked system): noise-free views give the true camera matrix, noisy ones the
    same estimate to 1e-9; two views suffice; one view, and views that leave the conic indefinite, take the fallback; NaN homographies are skipped."""
    from oracle import calibration_oracle as co

    hc.hc_zhang.restype = ctypes.c_int
    hc.hc_zhang.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    for (w, h), n, noise in (((1280, 1024), 100, 0.0), ((1280, 1024), 100, 0.3), ((640, 480), 7, 0.1), ((1080, 1920), 2, 0.0), ((800, 800), 25, 2.0)):
        p = synth.make_problem(2, n, seed=int(rng.integers(1 << 30)), noise=0.0)
        cam = p["true_cam"][1].copy()
        cam[4:6] = 0.0
        cam[6:] = 0.0
        cam[2], cam[3] = 0.5 * w + 7.0, 0.5 * h - 5.0
        uv = synth.project(cam[None], p["true_poses"] + np.r_[0, 0, 0, 0, 0, 400.0], p["obj"])[0] + rng.normal(0, noise, (n, len(p["obj"]), 2))
        H = np.ascontiguousarray(co.homographies(p["obj"][:, :2], uv))
        if n == 7:
            H[3] = np.nan
        K4 = np.zeros(4)
        used = hc.hc_zhang(len(H), P(H), float(w), float(h), P(K4))
        want = co.intrinsics_from_homographies(H, (w, h))
        assert used == 1, (w, h, n, noise)
        np.testing.assert_allclose(K4, [want[0, 0], want[1, 1], want[0, 2], want[1, 2]], rtol=1e-9)
        if noise == 0.0:
            np.testing.assert_allclose(K4, cam[:4], rtol=1e-7)
    K4 = np.zeros(4)
    assert hc.hc_zhang(1, P(H[:1].copy()), 1280.0, 1024.0, P(K4)) == 0 and np.array_equal(K4, [1280.0, 1280.0, 639.5, 511.5])
    junk = np.ascontiguousarray(rng.normal(size=(6, 3, 3)))
    used = hc.hc_zhang(6, P(junk), 1280.0, 1024.0, P(K4))
    want = co.intrinsics_from_homographies(junk, (1280, 1024))
    np.testing.assert_allclose(K4, [want[0, 0], want[1, 1], want[0, 2], want[1, 2]], rtol=1e-7)
    assert used == (0 if want[0, 0] == 1280.0 and want[0, 2] == 639.5 else 1)


def test_rotvec_from_matrix_is_the_references_formula(hc):
    from oracle import calibration_oracle as co

    rng = np.random.default_rng(4)
    for r in np.r_[rng.normal(0, 1.0, (20, 3)), np.zeros((1, 3)), [[1e-9, 0, 0]]]:
        R = np.ascontiguousarray(co.rodrigues(r))
        w = np.zeros(3)
        hc.hc_rotvec(P(R), P(w))
        want = np.nan_to_num(co.rodrigues_inv(R))   # (the reference's arccos is unclamped: NaN when rounding pushes the trace past 3 -- the kernels clamp)
        np.testing.assert_allclose(w, want, rtol=0, atol=1e-12)


def test_hostcheck_under_sanitizers():
    """The device math header and the LM decision compiled for the CPU with -fsanitize=address,undefined and driven through every
    test of this file in a child process (the ASan runtime has to be the first library of the process: LD_PRELOAD).  GPU
    AddressSanitizer is not available on the GPU boxes -- the CPU build is where out-of-bounds indexing of the packed triangles,
    the 6 x 6 helpers and the accumulator structs would show."""
    if os.environ.get("MCBA_HOSTCHECK_SANITIZE") == "1":
        pytest.skip("this IS the sanitizer run")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ, MCBA_HOSTCHECK_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__)], env=env, cwd=os.path.join(HERE, ".."), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
