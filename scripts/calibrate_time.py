"""Wall time of calibrate() (the initialiser that produces bundle_adjust's inputs; reference calibration.py:280-373) and of the pipeline
calibrate -> bundle_adjust, with the stages of calibrate() timed one by one.  Prints one JSON object.
usage: python scripts/calibrate_time.py [cameras,frames,rows,cols ...]   (default: the reference tutorial's 6,2130,5,7 and 6,10000,6,9)
MCBA_CAL_MISSING=<probability> drops whole detections."""
import contextlib
import functools
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m
from multicam_calibration_amd import calibration as cal

MISSING = float(os.environ.get("MCBA_CAL_MISSING", "0.1"))
REPS = int(os.environ.get("MCBA_CAL_REPS", "5"))
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(6, 2130, 5, 7), (6, 10000, 6, 9)]
acc = {}


def timed(name, fn):
    @functools.wraps(fn)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = acc.setdefault(name, [0.0, 0])
            d[0] += time.perf_counter() - t0
            d[1] += 1
    return w


from multicam_calibration_amd import ops

# the stages of calibrate(): the C-ABI crossings (ops.Problem methods) and the host-side pieces between them
for n in ("_sample_all_cameras", "_start_on_device", "_refine_intrinsics_on_device", "_spanning_tree", "_chain_extrinsics", "_pose_graph_on_device"):
    setattr(cal, n, timed(n, getattr(cal, n)))
for n in ("__init__", "calib_complete", "calib_start", "calib_homographies", "calib_view_poses", "view_subset", "calib_poses", "calib_graph", "calib_pairwise", "calib_consensus", "lm_run", "lm_result", "close"):
    setattr(ops.Problem, n, timed("ops." + n, getattr(ops.Problem, n)))

out = {}
for C, F, rows, cols in shapes:
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0, missing=MISSING)
    sizes = [(1280, 1024)] * C

    def run():
        np.random.seed(0)
        return cal.calibrate(p["uvs"], sizes, p["obj"], verbose=False)

    run()
    run()
    acc.clear()
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        ext, intr, poses, tree = run()
        ts.append(time.perf_counter() - t0)
    stages = {k: 1e3 * v[0] / REPS for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}
    # the call calibrate() feeds
    ok = ~np.isnan(poses).any(1)
    uv_ok, poses_ok = np.ascontiguousarray(p["uvs"][:, ok]), poses[ok]   # (frames no camera saw have no start pose: the reference's users drop them too)

    def ba():
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            return m.bundle_adjust(uv_ok, ext, intr, p["obj"], poses_ok, n_frames=int(ok.sum()), verbose=0, return_jac=False)

    ba()
    tb = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        r = ba()
        tb.append(time.perf_counter() - t0)
    out[f"{C}x{F}x{rows * cols}"] = {"calibrate_ms": 1e3 * float(np.median(ts)), "calibrate_ms_all": [1e3 * t for t in ts], "stages_ms": stages,
                                    "bundle_adjust_ms": 1e3 * float(np.median(tb)), "pipeline_ms": 1e3 * float(np.median(ts) + np.median(tb)),
                                    "ba_cost": float(r[4].cost), "ba_nfev": int(r[4].nfev), "missing": MISSING}
print(json.dumps(out, indent=1))
