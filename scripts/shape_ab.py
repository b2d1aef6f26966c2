import sys, time
sys.path.insert(0, ".")
import multicam_calibration_amd as m
for C, F, rows, cols in ((6, 1000, 6, 9), (6, 12500, 6, 9), (24, 6250, 10, 20), (6, 100000, 6, 9)):
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = m.ops.Problem(p["uvs"], p["obj"])
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
    lm.start(x0)
    for _ in range(60):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        K = 100
        for _ in range(K):
            lm.iterate(always_linearize=True)
        prob.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    print("C=%d F=%d N=%d: %.1f us/iter, cost %.10g" % (C, F, rows * cols, best * 1e6, lm.cost), flush=True)
    prob.close()
    del p
