"""Which seeded cases of tests/test_gpu_fuzz.py's sweep do not terminate?  Runs bundle_adjust() on the cases [lo, hi) with the sweep's settings
(ftol = xtol = 1e-13, gtol = 1e-11, max_nfev = 400) and prints one line per case that ends with status 0, plus a summary.  No oracle: seconds per
hundred cases.   usage: python tests/tools/fuzz_scan.py lo hi [extra bundle_adjust keywords as key=value ...]"""
import contextlib
import io
import json
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np

import multicam_calibration_amd as m
from test_gpu_fuzz import draw

lo, hi = int(sys.argv[1]), int(sys.argv[2])
extra = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    extra[k] = float(v) if v.replace(".", "").replace("e-", "").replace("-", "").isdigit() else v
bad, t0, nfev_all = [], time.time(), []
for it in range(lo, hi):
    mk, opts, fixed = draw(it)
    p = m.synth.make_problem(**mk)
    kw = dict(n_frames=None, ftol=1e-13, xtol=1e-13, gtol=1e-11, verbose=0, max_nfev=400, fix_intrinsics=fixed, return_jac=False, **opts)
    kw.update(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)[4]
    nfev_all.append(int(res.nfev))
    if res.status <= 0:
        h = res.lm.get("history", [])
        bad.append(dict(it=it, mk=mk, opts=opts, fixed=fixed, nfev=int(res.nfev), cost=float(res.cost), optimality=float(res.optimality), iterations=int(res.lm.get("iterations", -1))))
        print(json.dumps(bad[-1]), flush=True)
    if (it - lo) % 200 == 199:
        print(f"# {it - lo + 1} cases, {len(bad)} unterminated, {time.time() - t0:.0f}s", flush=True)
print(json.dumps({"range": [lo, hi], "unterminated": [b["it"] for b in bad], "total_nfev": int(np.sum(nfev_all)), "max_nfev": int(np.max(nfev_all)), "seconds": time.time() - t0}))
