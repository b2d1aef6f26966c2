"""How converged must the sampled views' start poses be?  mcba_calib_start with max_evaluations = 2 .. 60 per view: the crossing's time and the
evaluations + time of the joint refinement that follows, tutorial shape and 6 x 10 000 x 54."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import multicam_calibration_amd as m
from multicam_calibration_amd import calibration as cal, solver, ops

for arg in (sys.argv[1:] or ["6,2130,5,7", "6,10000,6,9", "2,50,6,9"]):
    C, F, rows, cols = (int(v) for v in arg.split(","))
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0, missing=0.1)
    prob = ops.Problem(p["uvs"], p["obj"], loss="linear")
    np.random.seed(0)
    views = cal._sample_all_cameras(prob.calib_complete(), 100)
    sizes = np.array([(1280, 1024)] * C, dtype=np.float64)
    free = np.tile(np.r_[np.ones(6, bool), np.zeros(6, bool)], C)
    sub = prob.view_subset(views, loss="linear")
    ref = None
    for ev in (60, 8, 4, 3, 2, 1):
        ts, tl = [], []
        for _ in range(7):
            t0 = time.perf_counter()
            k4, poses0 = prob.calib_start(views, sizes, max_evaluations=ev)
            t1 = time.perf_counter()
            cam0 = np.zeros((C, 12))
            cam0[:, :4] = k4
            res = solver.lm_solve(sub, np.concatenate([cam0.ravel(), poses0.ravel()]), ftol=1e-9, xtol=1e-9, gtol=1e-10, max_nfev=200, verbose=0, free_cam_mask=free)
            t2 = time.perf_counter()
            ts.append(t1 - t0); tl.append(t2 - t1)
        cam = res.x[: 12 * C].reshape(C, 12)
        ref = cam if ref is None else ref
        print(arg, "start evals", ev, "start ms %.3f" % (1e3 * np.median(ts)), "joint nfev", res.nfev, "status", res.status, "joint ms %.3f" % (1e3 * np.median(tl)),
              "sum %.3f" % (1e3 * (np.median(ts) + np.median(tl))), "cost %.10g" % res.cost, "intrinsics vs 60 (px) %.2e" % np.abs(cam[:, :4] - ref[:, :4]).max(), flush=True)
    sub.close(); prob.close()
