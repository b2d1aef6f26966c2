"""cProfile of calibrate()'s joint intrinsics refinement (host side of _refine_intrinsics_on_device) at the tutorial shape."""
import cProfile, pstats, sys
sys.path.insert(0, ".")
import numpy as np
import multicam_calibration_amd as m
from multicam_calibration_amd import calibration as cal, ops

p = m.synth.make_problem(6, 2130, rows=5, cols=7, seed=0, missing=0.1)
prob = ops.Problem(p["uvs"], p["obj"], loss="linear")
np.random.seed(0)
views = cal._sample_all_cameras(prob.calib_complete(), 100)
K0, poses0 = cal._start_on_device(prob, views, [(1280, 1024)] * 6)
for _ in range(3):
    cal._refine_intrinsics_on_device(prob, views, K0, poses0)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    cal._refine_intrinsics_on_device(prob, views, K0, poses0)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(35)
