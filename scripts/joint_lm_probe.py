"""How long the joint intrinsics refinement of calibrate() runs: evaluations, time per evaluation and the cost history, at a shape
(default: the tutorial's), at the tolerances calibrate() uses and at looser ones."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import multicam_calibration_amd as m
from multicam_calibration_amd import calibration as cal, solver, ops

for arg in (sys.argv[1:] or ["6,2130,5,7", "6,10000,6,9"]):
    C, F, rows, cols = (int(v) for v in arg.split(","))
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0, missing=0.1)
    prob = ops.Problem(p["uvs"], p["obj"], loss="linear")
    np.random.seed(0)
    complete = prob.calib_complete()
    views = cal._sample_all_cameras(complete, 100)
    K0, poses0 = cal._start_on_device(prob, views, [(1280, 1024)] * C)
    cam0 = np.zeros((C, 12))
    for c, K in enumerate(K0):
        cam0[c, :4] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    free = np.tile(np.r_[np.ones(6, bool), np.zeros(6, bool)], C)
    sub = prob.view_subset(views, loss="linear")
    x0 = np.concatenate([cam0.ravel(), poses0.ravel()])
    ref = None
    for tol in (1e-12, 1e-11, 1e-10, 1e-9, 1e-6):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            res = solver.lm_solve(sub, x0, ftol=tol, xtol=tol, gtol=1e-10, max_nfev=200, verbose=0, free_cam_mask=free)
            ts.append(time.perf_counter() - t0)
        cam = res.x[: 12 * C].reshape(C, 12)
        ref = cam if ref is None else ref
        print(arg, "tol", tol, "nfev", res.nfev, "status", res.status, "cost %.12g" % res.cost, "ms %.3f" % (1e3 * np.median(ts)), "us/eval %.1f" % (1e6 * np.median(ts) / res.nfev),
              "fx err", np.abs(cam[:, 0] - p["true_cam"][:, 0]).max().round(4), "intrinsics vs the 1e-12 run (px)", np.abs(cam[:, :4] - ref[:, :4]).max(), flush=True)
    sub.close()
    prob.close()
