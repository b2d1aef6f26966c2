"""The C-ABI library builds, loads without a GPU and exports every symbol include/mcba.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mcba.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcba_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from multicam_calibration_amd import build, ops

    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mcba.h but not exported by libmcba.so"
    bound = {s[0] for s in ops.SYMBOLS}
    assert bound == set(names), (sorted(bound - set(names)), sorted(set(names) - bound))
    assert ops.load_library().mcba_abi_version() == 7


def test_no_gpu_is_a_loud_error_not_a_fallback():
    """Without a GPU the product must raise (never route through the oracle / a CPU path)."""
    import numpy as np
    from multicam_calibration_amd import ops, synth

    n = ctypes.c_int()
    rc = ops.load_library().mcba_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    p = synth.make_problem(2, 3)
    with pytest.raises(ops.McbaError):
        ops.Problem(p["uvs"], p["obj"])
    # ... and so must calibrate()'s public pieces (round 6: pose graph and per-view starts are GPU kernels, not numpy)
    from multicam_calibration_amd import calibration as cal

    poses = np.zeros((2, 5, 6))
    for call in (lambda: cal.estimate_pairwise_camera_transform(poses[0], poses[1]), lambda: cal.consensus_calib_poses(poses, np.zeros((2, 6))),
                 lambda: cal.estimate_pose(p["uvs"][0], p["obj"], np.eye(3), np.zeros(5)), lambda: cal.calibrate(p["uvs"], [(640, 480)] * 2, p["obj"], verbose=False)):
        with pytest.raises(ops.McbaError):
            call()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multicam-calibration_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), fn


def test_measurement_scripts_do_not_touch_the_oracle():
    """scripts/ holds the measurement programs behind profiles/: none of them imports or runs anything under oracle/ (diagnostics that use it as
    their checker live in tests/tools/)."""
    sdir = os.path.join(ROOT, "scripts")
    for base, _, files in os.walk(sdir):
        for fn in files:
            if fn.endswith((".py", ".sh")):
                src = open(os.path.join(base, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|oracle/|oracle\.", src, flags=re.M), os.path.join(base, fn)


def test_more_than_40_cameras_is_refused_with_the_reason():
    """The reference has no camera limit; this build's is 40 per handle -- said in the exception a calibrate() / bundle_adjust() user of a bigger
    rig sees first (checked before any device is touched)."""
    import numpy as np
    from multicam_calibration_amd import ops

    with pytest.raises(ops.McbaError, match="at most 40"):
        ops.Problem(np.zeros((41, 3, 4, 2)), np.zeros((4, 3)))
