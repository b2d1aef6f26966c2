"""Per-kernel timing at other BASELINE shapes (development aid)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import multicam_calibration_amd as m

for (C, F, rows, cols, tag) in [(6, 1000, 6, 9, "config 2 size"), (6, 12500, 6, 9, "config 4 per-GPU shard"), (24, 6250, 10, 20, "config 5 per-GPU shard")]:
    t0 = time.perf_counter()
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    tg = time.perf_counter() - t0
    prob = m.ops.Problem(p["uvs"], p["obj"])
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0)
    lm.start(x0)
    for _ in range(5):
        lm.iterate()
    prob.profile_enable(True); prob.profile_read()
    for _ in range(10):
        lm.iterate()
    pr = prob.profile_read()
    prob.profile_enable(False)
    prob.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        lm.iterate()
    prob.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("== %s: C=%d F=%d N=%d (gen %.1fs)  %.1f us/iter = %.0f it/s ; cost %.6g accepted %d" % (tag, C, F, rows * cols, tg, dt * 1e6, 1 / dt, lm.cost, lm.iteration))
    print("   " + "  ".join("%s %.1f" % (k, 1e3 * ms / c) for k, (ms, c) in pr.items() if c))
    prob.close()
