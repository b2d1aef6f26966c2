"""Soak of tests/test_gpu_calibration.py::test_calibrate_random_rigs beyond its 40 committed cases: python tests/tools/calib_rig_soak.py 40 400"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import multicam_calibration_amd as mc
import test_gpu_calibration as t

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
calls = [0]
_ba = mc.bundle_adjust


def counted(*args, **kw):
    calls[0] += 1
    return _ba(*args, **kw)


mc.bundle_adjust = counted
for it in range(a, b):
    try:
        t.test_calibrate_random_rigs(mc, it)
    except BaseException as e:  # noqa: BLE001
        bad.append(it)
        print("case", it, t._draw_rig(it), "->", type(e).__name__, str(e)[:300].replace("\n", " "), flush=True)
print("cases", a, "..", b - 1, ":", len(bad), "failed", bad, "; downstream comparisons (pairs of bundle_adjust calls):", calls[0] // 2)
