"""Per-workgroup stamps of k_syrk inside the device-resident LM loop (development aid; needs a build with
MCBA_HIPCC_FLAGS=-DMCBA_SYRK_TIMING).   usage: [MCBA_SHAPE=C,F,rows,cols] python scripts/syrk_stamps.py [lib.so]
Round 6: the stamps (and the mcba_debug_* export this script reads) live in profiles/round6/patches/experiments_and_stamps.patch, not in the product sources:
`cd multicam-calibration_amd && git apply -p0 ../profiles/round6/patches/experiments_and_stamps.patch` first."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

if len(sys.argv) > 1:
    m.ops.LIB_PATH = sys.argv[1]
C, F, rows, cols = (int(a) for a in os.environ.get("MCBA_SHAPE", "6,10000,6,9").split(","))
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
lm.start(x0)
for _ in range(200 if C * F <= 60000 else 40):
    lm.iterate(always_linearize=True)
prob.synchronize()
out = np.zeros(12 + 12 * 512)
f = prob.lib.mcba_debug_syrk_stamps
f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
f.restype = ctypes.c_int
assert f(prob.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
G = int(out[0])
d = out[12:12 + 12 * G].reshape(G, 12)
names = ["entry -> trial scalars summed", "decision", "set-up", "V sums (loads)", "frame factors", "Y build", "MFMA", "barrier wait", "tile store"]
print("G = %d workgroups; shader cycles per workgroup, median / max:" % G)
for i, nm in enumerate(names):
    print("   %-32s %7.0f / %7.0f" % (nm, np.median(d[:, i]), d[:, i].max()))
w0, w1, xcc = d[:, 9], d[:, 10], d[:, 11].astype(int)
t0 = w0.min()
print("wall (us): workgroup start spread %.2f; duration min %.2f med %.2f max %.2f; first start -> last end %.2f; implied clock %.2f GHz"
      % ((w0.max() - t0) / 100, (w1 - w0).min() / 100, np.median(w1 - w0) / 100, (w1 - w0).max() / 100, (w1.max() - t0) / 100, np.median(d[:, :9].sum(1) / ((w1 - w0) * 10))))
print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
prob.close()
