import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m
p = m.synth.make_problem(6, 10000, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
prob.set_params(0, x0); prob.linearize(0)
for _ in range(5): prob.build_reduced(1e-3)
prob.synchronize()
out = np.zeros(8)
f = prob.lib.mcba_debug_syrk_stamps; f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]; f.restype = ctypes.c_int
f(prob.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
print("cycles: stage L + wait loads %d | Y build %d | MFMA %d | barrier %d | total %d" % tuple(out[:5]))
