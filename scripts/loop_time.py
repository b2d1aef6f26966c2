"""LM loop rate, device-resident solve vs host solve (development aid; bench.py is the contract).
usage: python scripts/loop_time.py C F [N_rows N_cols] """
import sys
import time

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rows, cols = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 9)
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
for mode, depth in (("host", 1), ("device", 1), ("device", 2), ("device", 4)):
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver=mode, depth=depth)
    lm.start(x0)
    for _ in range(30):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    dt = time.perf_counter() - t0
    print("%-6s depth %d: %.1f us/iter -> %.0f it/s ; cost %.12g accepted %d/%d lam %.3g" % (mode, depth, dt / K * 1e6, K / dt, lm.cost, lm.iteration, lm.steps, lm.lam), flush=True)
prob.profile_enable(True)
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
lm.start(x0)
for _ in range(50):
    lm.iterate()
prob.synchronize()
for k, (ms, n) in prob.profile_read().items():
    if n:
        print("%-18s %4d calls  %9.3f us avg" % (k, n, 1e3 * ms / n))
prob.close()
