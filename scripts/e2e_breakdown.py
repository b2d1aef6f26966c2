"""Where the wall time of the user-level call goes: bundle_adjust() from host arrays to the 5-tuple at BASELINE configs[2]
(6 x 10 000 x 54), return_jac=False, warm.  Every ops.Problem method (each one is a C-ABI crossing that synchronises when it
returns host data) and the host-side stages are wrapped with wall-clock timers; prints one JSON object.
MCBA_E2E_MISSING=<probability> drops whole detections (then result.fun / result.jac need the row mask: mcba_seen_bits).
usage: python scripts/e2e_breakdown.py [frames] [reps]"""
import contextlib
import functools
import io
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m
from multicam_calibration_amd import api, ops, solver

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
acc = {}
depth = [0]


def timed(name, fn):
    @functools.wraps(fn)
    def w(*a, **k):
        t0 = time.perf_counter()
        depth[0] += 1
        try:
            return fn(*a, **k)
        finally:
            depth[0] -= 1
            d = acc.setdefault(name, [0.0, 0])
            d[0] += time.perf_counter() - t0
            d[1] += 1
    return w


for name in dir(ops.Problem):
    if name.startswith("__") and name != "__init__":
        continue
    f = getattr(ops.Problem, name)
    if callable(f) and name not in ("_chk", "split_reduced", "_init_host_views"):
        setattr(ops.Problem, name, timed("ops." + name, f))
api.select_frames = timed("api.select_frames", api.select_frames)
solver.lm_solve = timed("solver.lm_solve", solver.lm_solve)
api.serialize_params = timed("api.serialize_params", api.serialize_params)
api.deserialize_params = timed("api.deserialize_params", api.deserialize_params)

import os

MISSING = float(os.environ.get("MCBA_E2E_MISSING", "0"))
CAMS, ROWS, COLS = 6, 6, 9
if os.environ.get("MCBA_E2E_SHAPE"):   # "cameras,frames,rows,cols" -- e.g. the reference tutorial's recording 6,2130,5,7
    CAMS, F, ROWS, COLS = (int(v) for v in os.environ["MCBA_E2E_SHAPE"].split(","))
p = m.synth.make_problem(CAMS, F, rows=ROWS, cols=COLS, seed=0, missing=MISSING)


if os.environ.get("MCBA_E2E_HOSTMASK") == "1":   # round 4 before mcba_seen_bits: the mask from a numpy pass over the caller's array
    ops.Problem.seen_bits = timed("host mask (numpy)", lambda self: np.packbits(~np.isnan(p["uvs"][:, np.arange(self.F)])))


def run():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        return api.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=F, verbose=0, return_jac=False)


run()  # cold: library load, first-touch allocations
run()
times = []
acc.clear()
for _ in range(REPS):
    t0 = time.perf_counter()
    out = run()
    times.append(time.perf_counter() - t0)
res = out[4]
tab = {k: {"ms_per_call_site": 1e3 * v[0] / REPS, "calls": v[1] / REPS} for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}
print(json.dumps({"workload": f"bundle_adjust {CAMS}x{F}x{ROWS * COLS} return_jac=False, warm, missing detections {MISSING}", "end_to_end_ms": {"min": 1e3 * min(times), "median": 1e3 * float(np.median(times)), "all": [1e3 * t for t in times]},
                  "nfev": int(res.nfev), "status": int(res.status), "cost": float(res.cost), "breakdown_ms (nested: inner calls are included in outer ones)": tab}, indent=1))
