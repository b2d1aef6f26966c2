import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m
p = m.synth.make_problem(6, 10000, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
prob.set_params(0, x0)
for _ in range(5): prob.linearize(0)
prob.synchronize()
out = np.zeros(3)
f = prob.lib.mcba_debug_gram_stamps; f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]; f.restype = ctypes.c_int
f(prob.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
print("k_gram wave (c=0, fb=1) cycles: prologue %d | point loop %d | epilogue %d" % tuple(out))
