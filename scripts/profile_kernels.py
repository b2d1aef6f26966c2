"""Fixed launch sequence for rocprofv3 counter passes: 12 LM steps + 6 Jacobian evaluations at 6x10kx54.
   rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 scripts/profile_kernels.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multicam_calibration_amd as m

C, F = 6, 10000
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0)
lm.start(x0)
for _ in range(12):
    lm.iterate(always_linearize=True)
for _ in range(6):
    prob.jacobian_eval(lm.cur, robust_scaled=True)
prob.synchronize()
print("done cost", lm.cost)
