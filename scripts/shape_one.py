"""One shape of scripts/shape_ab.py, for rocprofv3 --kernel-trace --stats:  python3 scripts/shape_one.py C F rows cols [iters]"""
import sys, time
sys.path.insert(0, ".")
import multicam_calibration_amd as m
C, F, rows, cols = (int(a) for a in sys.argv[1:5])
K = int(sys.argv[5]) if len(sys.argv) > 5 else 100
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
lm.start(x0)
for _ in range(40):
    lm.iterate(always_linearize=True)
prob.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    lm.iterate(always_linearize=True)
prob.synchronize()
print("C=%d F=%d N=%d: %.1f us/iter, cost %.10g" % (C, F, rows * cols, (time.perf_counter() - t0) / K * 1e6, lm.cost), flush=True)
prob.close()
