"""A/B of library builds on one box (development aid): k_gram time and the LM tick per library.
usage: python scripts/gram_ab.py lib1.so lib2.so ...   (each library runs in its own process)"""
import subprocess
import sys

CHILD = r'''
import sys, time
sys.path.insert(0, ".")
import multicam_calibration_amd as m
m.ops.LIB_PATH = sys.argv[1]
import ctypes
_l = ctypes.CDLL(sys.argv[1])
m.ops.SYMBOLS = [s for s in m.ops.SYMBOLS if hasattr(_l, s[0])]   # (an older build of the ABI: bind what it has)
if not hasattr(_l, "mcba_set_x_scale"):
    m.ops.Problem.set_x_scale = lambda self, x: None
if not hasattr(_l, "mcba_set_curvature_floor"):   # (a build from before the run-time curvature floor)
    del m.ops.Problem.set_curvature_floor
C, F = 6, 10000
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
lm.start(x0)
for _ in range(300):
    lm.iterate(always_linearize=True)
prob.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    best = min(best, (time.perf_counter() - t0) / 200)
prob.profile_enable(True)
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
lm.start(x0)
for _ in range(60):
    lm.iterate()
prob.synchronize()
prof = {k: 1e3 * ms / n for k, (ms, n) in prob.profile_read().items() if n}
print("%-40s tick %.1f us  cost %.12g | " % (sys.argv[1][-40:], best * 1e6, lm.cost) + "  ".join("%s %.1f" % (k, v) for k, v in prof.items()), flush=True)
prob.close()
'''
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
