"""k_solve_cam phase stamps INSIDE the LM loop (development aid; build with MCBA_HIPCC_FLAGS=-DMCBA_SOLVE_TIMING).
usage: [MCBA_LIB=lib.so] python scripts/solve_stamps_loop.py"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

if os.environ.get("MCBA_LIB"):
    m.ops.LIB_PATH = os.environ["MCBA_LIB"]
p = m.synth.make_problem(6, 10000, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
lm.start(x0)
acc = []
for i in range(300):
    lm.iterate(always_linearize=True)
    if i >= 200:
        acc.append(prob._auto_state[25:31].copy())
prob.synchronize()
a = np.median(np.array(acc), axis=0)
print("median stamps (cycles from kernel entry): state+loads %d | staged %d | factorised %d | swept %d | [diag %d, panel %d]" % tuple(a))
prob.close()
