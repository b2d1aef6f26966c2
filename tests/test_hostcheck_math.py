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
    hdrs = [os.path.join(HERE, "..", "multicam-calibration_amd", "csrc", h) for h in ("mcba_math.h", "mcba_lm.h", "mcba_lm_state.h")]
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
