"""Floor of one host<->GPU round trip through the ABI (tiny problem: kernels are ~free)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import multicam_calibration_amd as m
p = m.synth.make_problem(2, 64, rows=1, cols=2, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
prob.set_params(0, x0)
prob.linearize(0)
for name, fn in [("cost (2 kernels + D2H + sync)", lambda: prob.cost(0)),
                 ("reduce_fetch (3 kernels + D2H + sync)", lambda: prob.reduce_fetch(1e-3, 0)),
                 ("synchronize only", lambda: prob.synchronize()),
                 ("linearize (launch only, no sync)", lambda: prob.linearize(0))]:
    for _ in range(20): fn()
    prob.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): fn()
    prob.synchronize()
    print("%-40s %.1f us" % (name, (time.perf_counter() - t0) / 500 * 1e6))
red = prob.reduce_fetch(1e-3, 0)
dc = np.zeros(prob.n)
for _ in range(20): prob.step_fetch(dc, 1e-3, 0, 1, True)
t0 = time.perf_counter()
for _ in range(500): prob.step_fetch(dc, 1e-3, 0, 1, True)
print("%-40s %.1f us" % ("step_fetch (H2D + 3 kernels + D2H + sync)", (time.perf_counter() - t0) / 500 * 1e6))
