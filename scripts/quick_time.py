"""Quick per-kernel timing on one GPU (development aid; bench.py is the contract)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
t0 = time.perf_counter()
prob = m.ops.Problem(p["uvs"], p["obj"])
print("create+upload %.1f ms" % ((time.perf_counter() - t0) * 1e3))
prob.set_params(0, x0)
prob.profile_enable(True)
for it in range(12):
    prob.linearize(0)
    prob.build_reduced(1e-3)
    red = prob.get_reduced()
    dc = np.linalg.solve(red["S0"] + 1e-3 * np.diag(red["diagU"]), red["rhs"])
    prob.step(dc, 1e-3, 0, 1)
    t = prob.get_trial()
    if it == 1:
        prob.profile_read()
prob.jacobian_eval(0, False)
prob.jacobian_eval(0, True)
prof = prob.profile_read()
for k, (ms, n) in prof.items():
    if n:
        print("%-18s %4d calls  %9.3f us avg" % (k, n, 1e3 * ms / n))
print("cost", red["scal"][0], "trial", t[:5])
prob.profile_enable(False)
t0 = time.perf_counter()
res = m.solver.lm_solve(prob, x0, ftol=1e-10, xtol=1e-12, gtol=1e-8, verbose=2, max_nfev=60)
dt = time.perf_counter() - t0
print("lm_solve: %d iterations, nfev %d in %.3f s -> %.1f it/s; status %d cost %.10g opt %.2e" % (res.lm["iterations"], res.nfev, dt, res.lm["steps"] / dt, res.status, res.cost, res.optimality))
