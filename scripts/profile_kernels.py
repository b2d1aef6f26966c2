"""Fixed launch sequence for rocprofv3 counter passes: 12 LM steps + 6 Jacobian evaluations at 6x10kx54
   (MCBA_SHAPE="C,F,rows,cols", e.g. "24,6250,10,20": another shape, reduced system solved on the GPU, no Jacobian evaluations;
   MCBA_FIXED=1: intrinsics held fixed).
   rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 scripts/profile_kernels.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multicam_calibration_amd as m

C, F, rows, cols = 6, 10000, 6, 9
shape = os.environ.get("MCBA_SHAPE")
if shape:
    C, F, rows, cols = (int(v) for v in shape.split(","))
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
if os.environ.get("MCBA_FIXED"):   # BASELINE configs[1]: intrinsics held fixed (camera block 6 wide)
    assert prob.set_camera_block(6)
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, **(dict(reduced_solver="device", depth=2) if shape else {}))
lm.start(x0)
for _ in range(12):
    lm.iterate(always_linearize=True)
for _ in range(0 if shape else 6):
    prob.jacobian_eval(lm.cur, robust_scaled=True)
prob.synchronize()
print("done cost", lm.cost)
