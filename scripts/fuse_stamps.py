"""Wall-clock stamps of k_solve_backsub (development aid; build with MCBA_HIPCC_FLAGS=-DMCBA_FUSE_TIMING): the solve workgroup
and back-substitution workgroup 1, 10 ns ticks.   usage: [MCBA_LIB=lib.so] python scripts/fuse_stamps.py
Round 6: the stamps (and the mcba_debug_* export this script reads) live in profiles/round6/patches/experiments_and_stamps.patch, not in the product sources:
`cd multicam-calibration_amd && git apply -p0 ../profiles/round6/patches/experiments_and_stamps.patch` first."""
import ctypes
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
for i in range(200):
    lm.iterate(always_linearize=True)
prob.synchronize()
out = np.zeros(8)
f = prob.lib.mcba_debug_fuse_stamps
f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
f.restype = ctypes.c_int
assert f(prob.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
t0 = out[2]
us = lambda v: (v - t0) / 100.0
print("us from the back-substitution workgroup's entry: camera step fetched %.2f | products start %.2f | frame steps ready %.2f | end %.2f || the solve releases its final word %.2f"
      % (us(out[3]), us(out[4]), us(out[5]), us(out[6]), us(out[7])))
prob.close()
