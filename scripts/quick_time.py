"""Quick per-kernel timing on one GPU (development aid; bench.py is the contract)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"]
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
for mode in modes:
    os.environ["MCBA_GRAM_SPLIT"] = mode
    prob = m.ops.Problem(p["uvs"], p["obj"])
    prob.set_params(0, x0)
    prob.profile_enable(True)
    for it in range(12):
        prob.linearize(0)
        prob.build_reduced(1e-3)
        red = prob.get_reduced()
        dc = np.linalg.solve(red["S0"] + 1e-3 * np.diag(red["diagU"]), red["rhs"])
        prob.step_linearize(dc, 1e-3, 0, 1)
        t = prob.get_trial()
        if it == 1:
            prob.profile_read()
    prob.jacobian_eval(0, False)
    prob.jacobian_eval(0, True)
    prof = prob.profile_read()
    print("== MCBA_GRAM_SPLIT=%s" % mode)
    for k, (ms, n) in prof.items():
        if n:
            print("%-18s %4d calls  %9.3f us avg" % (k, n, 1e3 * ms / n))
    print("cost %.12g trial %s" % (red["scal"][0], t[:3]))
    prob.profile_enable(False)
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0)
    lm.start(x0)
    for _ in range(20):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    dt = time.perf_counter() - t0
    print("LM loop: %.1f us/iter -> %.0f it/s ; cost %.12g" % (dt / 200 * 1e6, 200 / dt, lm.cost))
    prob.close()
