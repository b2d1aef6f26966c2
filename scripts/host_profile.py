"""Where does the host time of one LM iteration go?  (development aid)"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = 6, 10000
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0)
lm.start(x0)
for _ in range(20):
    lm.iterate(always_linearize=True)
prob.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    lm.iterate(always_linearize=True)
prob.synchronize()
print("plain: %.1f us/iter" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    lm.iterate(always_linearize=True)
pr.disable()
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(18)
